#!/bin/bash
# single SQ-counter pass (safe set) for a command; prints per-kernel averages. usage: tools/pmc_sq.sh <outdir> <filter> -- cmd...
out=$1; filt=$2; shift 3
mkdir -p $out; export TMPDIR=/tmp
i=0
for p in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  ( cd /tmp && timeout 120 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/p$i -o p -- "$@" > $GRAFT_REPO_ROOT/$out/p$i.log 2>&1 )
  i=$((i+1))
done
python - "$out" "$filt" <<'PY'
import csv, glob, collections, sys
out, filt = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if not any(x in k for x in filt.split('|')): continue
        k = k[:80]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/p0/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if any(x in k for x in filt.split('|')): dur[k[:80]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    d = dur.get(k, [0])
    print(k, f"| dispatches {len(d)} avg {sum(d)/max(len(d),1):.1f} us")
    for c in sorted(v): print(f"    {c:30s} {v[c] / max(cnt[(k,c)],1):.5g}")
PY
