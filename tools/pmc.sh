#!/bin/bash
# Collect hardware counters for a command in several rocprofv3 passes (PMC only + kernel trace) and print a per-kernel table.
# usage: tools/pmc.sh <outdir> <kernel-substring> -- <command...>
# (a pass with TA_* counters next to GRBM/TD counters aborted rocprofv3 and hung a box for 25 minutes: left out on purpose)
out=$1; filt=$2; shift 3
mkdir -p $out
export TMPDIR=/tmp
declare -a PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE"
 "GRBM_GUI_ACTIVE"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"
 "TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
 "TCC_EA0_RDREQ_sum TCC_BUSY_sum"
)
i=0
for p in "${PASSES[@]}"; do
  ( cd /tmp && timeout 90 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/p$i -o p -- "$@" > $GRAFT_REPO_ROOT/$out/p$i.log 2>&1 )
  i=$((i+1))
done
python - "$out" "$filt" <<'PY'
import csv, glob, collections, sys
out, filt = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if filt not in k: continue
        k = k[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/p0/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if filt in k: dur[k[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    d = dur.get(k, [0])
    print(k, f"| dispatches {len(d)} avg {sum(d)/max(len(d),1):.1f} us")
    for c in sorted(v): print(f"    {c:36s} {v[c] / max(cnt[(k,c)],1):.5g}")
PY
