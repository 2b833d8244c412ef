# kernel lab for the two-pass SpMV on C3: kernel times per ablation (results are wrong when ablate != 0)
# ablate bits 0-1: pass 2 (1 stream only, 2 no uniform-row fold), bits 2-4: pass 1 (4 no store, 8 no LDS gather, 12 plain store, 16 no run sums)
# EXTRA: further configs.py flags, e.g. EXTRA='--opt spmv_pb_fold=-1'
cd /tmp && export TMPDIR=/tmp
for ab in ${ABS:-0 1 2 4 8}; do
  rm -rf /tmp/pbl; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pbl -o c3 -- python3 $GRAFT_REPO_ROOT/bench/configs.py c3 --steps 10 --ablate $ab --opt spmv_2d=-1 $EXTRA > /tmp/pbl.log 2>&1
  echo "ablate=$ab"; grep "^{" /tmp/pbl.log; python3 $GRAFT_REPO_ROOT/bench/kstats.py /tmp/pbl/c3_kernel_stats.csv pb_pass pb_xperm
done
