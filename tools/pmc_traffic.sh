#!/bin/bash
# HBM-side traffic per kernel launch of the bench workload, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE
# in separate rocprofv3 --pmc passes (kernel trace only), FETCH_SIZE doubled on gfx950 (128-B requests tallied at 64 B);
# both counters are in KiB.  Infinity-Cache hits are included (memory-side L2 requests).  Writes profiles JSON to $1.
out=${1:-gpurun_out/pmc_traffic.json}
mkdir -p gpurun_out/pmct; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmct/$c -o p -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-matched-recall --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/pmct/$c.log 2>&1 )
done
python - "$out" <<'PY'
import csv, glob, collections, json, sys
vals = collections.defaultdict(lambda: collections.defaultdict(dict))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmct/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                vals[r["Kernel_Name"]][c].setdefault(len(vals[r["Kernel_Name"]][c]), float(r["Counter_Value"]))
res = []
for k, v in vals.items():
    if not any(x in k for x in ("gemm_", "attention_kernel", "layernorm")): continue
    f = [v["FETCH_SIZE"][i] for i in sorted(v["FETCH_SIZE"])]
    w = [v["WRITE_SIZE"][i] for i in sorted(v["WRITE_SIZE"])]
    n = min(len(f), len(w))
    # one kernel name serves several problem sizes (proj / fc2 / text tower): cluster launches by fetched bytes (+-15 %)
    clusters = []
    for i in sorted(range(n), key=lambda i: f[i]):
        if clusters and f[i] <= clusters[-1]["lo"] * 1.15: clusters[-1]["ids"].append(i)
        else: clusters.append({"lo": max(f[i], 1.0), "ids": [i]})
    for cl in clusters:
        ids = cl["ids"]
        fetch = 2.0 * 1024.0 * sum(f[i] for i in ids) / len(ids)        # KiB -> B, x2 gfx950 correction
        write = 1024.0 * sum(w[i] for i in ids) / len(ids)
        res.append({"kernel": k[:100], "launches": len(ids), "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                    "traffic_bytes_per_launch": fetch + write})
res.sort(key=lambda e: -e["traffic_bytes_per_launch"] * e["launches"])
json.dump({"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on `bench.py --steps 1 --warmup 0`; KiB; FETCH_SIZE x2 (gfx950); "
                     "memory-side L2 requests, Infinity-Cache hits included; launches of one kernel clustered by fetched bytes",
           "kernels": res}, open(sys.argv[1], "w"), indent=1)
for e in res[:14]: print(e["kernel"][:64], "launches", e["launches"], "fetch %.2f GB write %.2f GB" % (e["fetch_bytes_per_launch"] / 1e9, e["write_bytes_per_launch"] / 1e9))
PY
