# Round-end evidence: bench line, rocprof kernel stats of the same command, all-config table, PMC passes.
# Usage (on the GPU box): bash bench/round_profile.sh <tag>   -> files under gpurun_out/<tag>_*
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT && timeout -k 10 500 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -5 $OUT/${TAG}_bench.err; exit 1; }
cat $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/${TAG}_prof.log 2>&1 || { tail -5 $OUT/${TAG}_prof.log; exit 1; }
python3 $GRAFT_REPO_ROOT/bench/kstats.py $OUT/${TAG}_prof/bench_kernel_stats.csv spmv pb_ | head -8
cd $GRAFT_REPO_ROOT && timeout -k 10 600 python bench/configs.py c1 c2 c3 c4 c5r c5x c5 > $OUT/${TAG}_configs.log 2> $OUT/${TAG}_configs.err || { tail -5 $OUT/${TAG}_configs.err; exit 1; }
grep "^{" $OUT/${TAG}_configs.log > $OUT/${TAG}_configs.jsonl; cat $OUT/${TAG}_configs.jsonl
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_cfg -o configs -- python3 $GRAFT_REPO_ROOT/bench/configs.py c1 c2 c3 c4 c5r --steps 10 > $OUT/${TAG}_prof_cfg.log 2>&1 || { tail -5 $OUT/${TAG}_prof_cfg.log; exit 1; }
cd $GRAFT_REPO_ROOT && bash bench/pmc_traffic.sh > $OUT/${TAG}_pmc_traffic.log 2>&1 || { tail -5 $OUT/${TAG}_pmc_traffic.log; exit 1; }
CFG=c3 KFILTER=pb_pass,pb_xperm PASSES="0 1 2 3" bash bench/pmc_cfg.sh > $OUT/${TAG}_pmc_two_pass_c3.txt 2>&1 || { tail -5 $OUT/${TAG}_pmc_two_pass_c3.txt; exit 1; }
CFG=c5r KFILTER=pb_pass,pb_xperm PASSES="0 1 2" bash bench/pmc_cfg.sh > $OUT/${TAG}_pmc_two_pass_c5r.txt 2>&1 || { tail -5 $OUT/${TAG}_pmc_two_pass_c5r.txt; exit 1; }
echo PROFILE_DONE
