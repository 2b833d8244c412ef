# One-sweep pass lab: C2 assembly through the plain LSD sort with parts of rs_onesweep switched off
# (ablate 0x4000 contiguous stores, 0x6000 contiguous stores and no match-any; the result is then an unsorted
# permutation of valid keys, so everything downstream stays in range.  A switch that skipped the stores left
# uninitialised keys to merge_emit and faulted the GPU: not offered).
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for AB in ${ABS:-0 16384 24576}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sweep_lab_$AB -o t -- python3 $GRAFT_REPO_ROOT/bench/configs.py ${CFG:-c2} --steps 3 --asm-only --opt sort_msd=-1 --ablate $AB > $OUT/sweep_lab_$AB.log 2>&1 || { tail -5 $OUT/sweep_lab_$AB.log; exit 1; }
  echo "ablate=$AB"; python3 $GRAFT_REPO_ROOT/bench/kstats.py $OUT/sweep_lab_$AB/t_kernel_stats.csv rs_onesweep 2>/dev/null
done
