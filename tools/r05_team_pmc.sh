#!/bin/bash
# Round 5: memory-side fetch / write of the proj / fc2 GEMMs with the default walk and with the team walk (debug bit 18):
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (kernel trace only), KiB, FETCH_SIZE x 2 on gfx950.
out=gpurun_out/team; mkdir -p $out; export TMPDIR=/tmp
shapes="${SHAPES:-proj_stats2 fc2_stats2}"
for d in ${DBGS:-0 262144}; do for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_${d}_$c -o p -- \
      python $GRAFT_REPO_ROOT/tools/gemm_bench.py --variants 0 --dbg $d --shapes $shapes --iters 3 --warmup 2 > /dev/null 2>&1 )
done; done
python - <<'PY'
import csv, glob, collections, os
out = "gpurun_out/team"
rows = []
for d in os.environ.get("DBGS", "0 262144").split():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{out}/pmc_{d}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and "gemm_" in r["Kernel_Name"]:
                    acc[r["Kernel_Name"][:60]][c].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        # launches alternate shapes in run order: cluster by fetched bytes
        f = sorted(v["FETCH_SIZE"]); w = sorted(v["WRITE_SIZE"])
        half = len(f) // 2
        for name, sl in (("smaller-K shape", slice(0, half)), ("larger-K shape", slice(half, None))):
            ff, ww = f[sl], w[sl]
            if ff: rows.append(f"dbg {d:>7s} {k:60s} {name:16s} launches {len(ff):2d}  fetch {2 * 1024 * sum(ff) / len(ff) / 1e9:6.2f} GB  write {1024 * sum(ww) / max(len(ww), 1) / 1e9:6.2f} GB")
open(f"{out}/pmc_ab.txt", "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
rm -rf $out/pmc_*_FETCH_SIZE $out/pmc_*_WRITE_SIZE
