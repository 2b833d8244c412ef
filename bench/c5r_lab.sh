# C5 row shard two-pass SpMV: per-kernel times (rocprofv3 kernel stats) of bench/configs.py c5r.  On the GPU box:
#   bash bench/c5r_lab.sh <tag> [extra configs.py args]   -> gpurun_out/<tag>_c5r.jsonl, <tag>_c5r_kernels.txt
TAG=${1:-lab}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/c5rlab
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5rlab -o c5r -- python3 $GRAFT_REPO_ROOT/bench/configs.py c5r --steps 20 "$@" > $OUT/${TAG}_c5r.log 2>&1 || { tail -5 $OUT/${TAG}_c5r.log; exit 1; }
grep "^{" $OUT/${TAG}_c5r.log > $OUT/${TAG}_c5r.jsonl
python3 $GRAFT_REPO_ROOT/bench/kstats.py /tmp/c5rlab/c5r_kernel_stats.csv pb_pass pb_xperm | tee $OUT/${TAG}_c5r_kernels.txt
