#!/bin/bash
# one rocprofv3 --pmc pass with the given counters; per-kernel averages. usage: tools/pmc_custom.sh "<counters>" <filter> -- cmd...
ctr=$1; filt=$2; shift 3
rm -rf gpurun_out/pmcc; mkdir -p gpurun_out/pmcc; export TMPDIR=/tmp
( cd /tmp && timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcc -o p -- "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmcc.log 2>&1 )
python - "$filt" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(x in r["Kernel_Name"] for x in sys.argv[1].split("|")): agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()): print(f"    {c:32s} {sum(x)/len(x):.5g}  (n={len(x)})")
PY
rm -rf gpurun_out/pmcc
