# In-LDS sort lab: C2 assembly with the number of in-LDS passes capped (ablate = passes + 1; results are wrong then).
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for AB in ${ABS:-0 1 2 4}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/msd_lab_$AB -o t -- python3 $GRAFT_REPO_ROOT/bench/configs.py ${CFG:-c2} --steps 3 --asm-only --ablate $AB "$@" > $OUT/msd_lab_$AB.log 2>&1 || { tail -5 $OUT/msd_lab_$AB.log; exit 1; }
  echo "ablate=$AB"; python3 $GRAFT_REPO_ROOT/bench/kstats.py $OUT/msd_lab_$AB/t_kernel_stats.csv rs_local msd_ 2>/dev/null
done
