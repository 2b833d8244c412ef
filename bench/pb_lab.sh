# kernel-lab: ablations of the two-pass SpMV on C3 (results are wrong when ablate != 0)
cd /tmp && export TMPDIR=/tmp
for ab in 0 1 2 4 8; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pb_lab_$ab -o c3 -- python3 $GRAFT_REPO_ROOT/bench/configs.py c3 --steps 5 --ablate $ab > $GRAFT_REPO_ROOT/gpurun_out/pb_lab_$ab.log 2>&1 || exit 1
  echo "ablate=$ab"; grep -E "pb_pass" $GRAFT_REPO_ROOT/gpurun_out/pb_lab_$ab/c3_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,150-
done
