OUT=$GRAFT_REPO_ROOT/gpurun_out
pp() { python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['config'][:30], 'spmv_ms', j['spmv_ms'], 'fold', j.get('pb_fold'))
"; }
for f in 0 -1; do
  echo "== un-profiled fold=$f"
  timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench/configs.py c5 --steps 30 --opt spmv_pb_fold=$f 2>&1 | pp || exit 1
  echo "== profiled fold=$f"
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5ab_$f -o t -- python3 $GRAFT_REPO_ROOT/bench/configs.py c5 --steps 30 --opt spmv_pb_fold=$f 2>&1 | pp) || exit 1
  python3 $GRAFT_REPO_ROOT/bench/kstats.py $OUT/c5ab_$f/t_kernel_stats.csv pb_pass 2>/dev/null
done
