# C3 two-pass SpMV: per-kernel times (rocprofv3 kernel stats) of bench/configs.py c3.  On the GPU box:
#   bash bench/c3_lab.sh <tag> [extra configs.py args]   -> gpurun_out/<tag>_c3.jsonl, <tag>_c3_kernels.txt
TAG=${1:-lab}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/c3lab
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3lab -o c3 -- python3 $GRAFT_REPO_ROOT/bench/configs.py c3 --steps 20 --opt spmv_eager=1 "$@" > $OUT/${TAG}_c3.log 2>&1 || { tail -5 $OUT/${TAG}_c3.log; exit 1; }
grep "^{" $OUT/${TAG}_c3.log > $OUT/${TAG}_c3.jsonl; cat $OUT/${TAG}_c3.jsonl
python3 $GRAFT_REPO_ROOT/bench/kstats.py /tmp/c3lab/c3_kernel_stats.csv pb_pass pb_xperm | tee $OUT/${TAG}_c3_kernels.txt
