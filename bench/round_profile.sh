# Round-end evidence: bench line, rocprof kernel stats of the same command, all-config table.
# Usage (on the GPU box): bash bench/round_profile.sh <tag>   -> files under gpurun_out/<tag>_*
TAG=${1:-r01_e}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT && timeout -k 10 400 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -5 $OUT/${TAG}_bench.err; exit 1; }
cat $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/${TAG}_prof.log 2>&1 || { tail -5 $OUT/${TAG}_prof.log; exit 1; }
head -6 $OUT/${TAG}_prof/bench_kernel_stats.csv | cut -c1-160
cd $GRAFT_REPO_ROOT && timeout -k 10 600 python bench/configs.py c1 c2 c3 c4 c5r > $OUT/${TAG}_configs.log 2> $OUT/${TAG}_configs.err || { tail -5 $OUT/${TAG}_configs.err; exit 1; }
cat $OUT/${TAG}_configs.log
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_cfg -o configs -- python3 $GRAFT_REPO_ROOT/bench/configs.py c1 c2 c3 c4 c5r --steps 10 > $OUT/${TAG}_prof_cfg.log 2>&1 || { tail -5 $OUT/${TAG}_prof_cfg.log; exit 1; }
