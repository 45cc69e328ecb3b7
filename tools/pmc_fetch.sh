#!/bin/bash
# FETCH_SIZE (x2 gfx950 correction, bytes) per launch of kernels matching $1 for the command after --
filt=$1; shift 2
mkdir -p gpurun_out/pmcf; rm -rf gpurun_out/pmcf/*; export TMPDIR=/tmp
( cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcf -o p -- "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmcf.log 2>&1 )
python - "$filt" <<'PY'
import csv, glob, sys, collections
v = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and sys.argv[1] in r["Kernel_Name"]: v[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]) * 2048 / 1e9)
for k, x in v.items(): print(k, "launches", len(x), "fetch GB per launch: min %.2f median %.2f max %.2f" % (min(x), sorted(x)[len(x)//2], max(x)))
PY
