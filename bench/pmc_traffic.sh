# HBM traffic of the metric's kernel (spmv_stream_kernel on the C2 workload), as MI355X_MICROARCH.md prescribes:
# one rocprofv3 --pmc pass per counter, gfx950 correction for FETCH_SIZE.  Run on the GPU box:
#   bash bench/pmc_traffic.sh        -> gpurun_out/pmc_traffic.json  (copy to profiles/pmc_traffic.json)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
mkdir -p $OUT
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-c5 > $OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $OUT/$tag.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv, json, sys, collections, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
# the metric's kernel = the spmv_stream_kernel instantiation launched most often (the timed loop's); the other
# legs of bench.py launch other instantiations a few times (first products of their records) and are left out
names = collections.Counter(row["Kernel_Name"].split("(")[0]
                            for row in csv.DictReader(open(os.path.join(out, "FETCH_SIZE", "p_counter_collection.csv")))
                            if "spmv_stream_kernel" in row["Kernel_Name"])
kname = names.most_common(1)[0][0]
for tag in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
    for row in csv.DictReader(open(os.path.join(out, tag, "p_counter_collection.csv"))):
        if row["Kernel_Name"].split("(")[0] != kname:
            continue
        a = acc[row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
m = {k: v[0] / v[1] for k, v in acc.items()}
read_b = 2 * m["FETCH_SIZE"] * 1024          # gfx950: 128-B read requests are tallied at 64 B
write_b = m["WRITE_SIZE"] * 1024
sha_path = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), ".build_sha")
sha = open(sha_path).read().split() if os.path.exists(sha_path) else ["unknown", "0"]
res = {
    "git_sha": sha[0] + (" + %s uncommitted files" % sha[1] if len(sha) > 1 and sha[1] != "0" else ""),
    "command": "bench/pmc_traffic.sh: rocprofv3 --pmc <counter> --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-c5 (one pass per counter group)",
    "kernel": kname + " on the C2 workload (5-point Laplacian 4096^2, f64)",
    "dispatches_averaged": acc["FETCH_SIZE"][1],
    "FETCH_SIZE_KB_raw": m["FETCH_SIZE"], "WRITE_SIZE_KB": m["WRITE_SIZE"],
    "correction": "gfx950: FETCH_SIZE counts 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM) -> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE exact",
    "read_bytes": read_b, "write_bytes": write_b,
    "spmv_c2_f64_hbm_bytes_per_launch": int(read_b + write_b),
    "algorithmic_bytes": 83869696 * 10 + 81904 * 4 + 16777216 * 18,
    "TCC_HIT_sum": m.get("TCC_HIT_sum"), "TCC_MISS_sum": m.get("TCC_MISS_sum"),
}
json.dump(res, open(os.path.join(os.path.dirname(out), "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res))
PY
