# one counter group per run (rocprofv3 does not split passes); summaries land in gpurun_out/pmc_${CFG:-c4}/
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${CFG:-c4}
mkdir -p $OUT
CGRP=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum")
for i in ${PASSES:-0 1 2 3 4}; do
  grp=${CGRP[$i]}
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o p -- python3 $GRAFT_REPO_ROOT/bench/configs.py ${CFG:-c4} --steps 10 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
  python3 - "$OUT/p$i/p_counter_collection.csv" <<'PY'
import csv, sys, collections, os
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if not any(f in k for f in os.environ.get("KFILTER", "spmm").split(",")): continue
    key = (k.split("(")[0][-60:], row["Counter_Name"])
    acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print("%-62s %-28s %14.1f per dispatch (%d)" % (k, c, v / n, n))
PY
done
