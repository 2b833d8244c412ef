# Per-kernel timeline of the assembly of one config (kernel trace of bench/configs.py <cfg>).
# Usage (GPU box): CFG=c2 bash bench/asm_profile.sh [extra configs.py flags]
CFG=${CFG:-c2}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/asm_${CFG} -o t -- python3 $GRAFT_REPO_ROOT/bench/configs.py $CFG --steps 5 "$@" > $OUT/asm_${CFG}.log 2>&1 || { tail -5 $OUT/asm_${CFG}.log; exit 1; }
grep "^{" $OUT/asm_${CFG}.log
python3 $GRAFT_REPO_ROOT/bench/kstats.py $OUT/asm_${CFG}/t_kernel_stats.csv | head -40
