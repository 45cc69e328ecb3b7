#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for prec in bf16x3 fp32; do
  rm -rf /tmp/jp_$prec
  JOINT_PRECISION=$prec rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/jp_$prec -o jp -- python $R/tools/joint_profile.py moment_segmentation > /tmp/jp_$prec.log 2>&1
  grep "videos/s" /tmp/jp_$prec.log | tail -1
  f=$(find /tmp/jp_$prec -name "*kernel_stats.csv" | head -1)
  echo "== $prec kernel stats ($f)"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f'{float(r["TotalDurationNs"]) / tot * 100:5.1f} %  calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  {r["Name"][:110]}')
print("total kernel time per batch (21 batches):", tot / 21 / 1e6, "ms")
PY
  cp "$f" $R/gpurun_out/joint_seg_${prec}_kernel_stats.csv
done
