# Counter passes of one config of bench/configs.py, one rocprofv3 run per group (rocprofv3 does not split passes).
# Usage (on the GPU box): CFG=c3 KFILTER=pb_pass1 PGROUPS="A B C;D E" [ARGS="--opt k=v"] bash bench/pmc_any.sh <tag>
#   -> gpurun_out/<tag>_pmc.txt (per kernel and counter: mean per dispatch)
TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
: > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc.txt
IFS=';' read -ra GR <<< "$PGROUPS"
i=0
for grp in "${GR[@]}"; do
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o p -- python3 $GRAFT_REPO_ROOT/bench/configs.py ${CFG:-c3} --steps 10 $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -5 $OUT/p$i.log; exit 1; }
  python3 - "$OUT/p$i/p_counter_collection.csv" >> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc.txt <<'PY'
import csv, sys, collections, os
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if not any(f in k for f in os.environ.get("KFILTER", "pb_pass").split(",")): continue
    key = (k.split("(")[0][-60:], row["Counter_Name"])
    acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print("%-62s %-36s %16.1f per dispatch (%d)" % (k, c, v / n, n))
PY
  i=$((i+1))
done
cat $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc.txt
