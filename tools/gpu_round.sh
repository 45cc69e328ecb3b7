#!/bin/bash
# One GPU-box trip. Outputs under gpurun_out/.  Usage: bash tools/gpu_round.sh [tests] [bench] [ab] [prof] [probe]
mkdir -p gpurun_out
export TMPDIR=/tmp
for what in "$@"; do
case $what in
probe)
  ( cd tools/probes && for f in *.hip; do hipcc --offload-arch=gfx950 -O2 $f -o /tmp/${f%.hip} 2>/dev/null && /tmp/${f%.hip}; done ) > gpurun_out/probes.log 2>&1 ;;
tests)
  timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
  tail -4 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log ;;
ab)
  for k in 1 2; do timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --gemm-kernel $k > gpurun_out/bench_k$k.log 2>&1; done
  python - <<'PY'
import json
for k in (1,2):
    try:
        d=json.loads(open(f"gpurun_out/bench_k{k}.log").read().strip().splitlines()[-1])
        print("gemm-kernel",k,"frames/s %.1f"%d["value"])
        for e in d["roofline"]["breakdown"][:7]: print("   ",e["kernel"],e["tag"],e["dims"],"avg_ms %.3f"%e["avg_ms"],"share %.3f"%e["share"],"TF %.0f"%e.get("tflops",0))
    except Exception as ex: print(k,"failed",ex); print(open(f"gpurun_out/bench_k{k}.log").read()[-1500:])
PY
  ;;
bench)
  timeout 900 python bench.py > gpurun_out/bench.log 2>&1; tail -c 2500 gpurun_out/bench.log ;;
prof)
  cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; cd $GRAFT_REPO_ROOT
  ls -R gpurun_out/prof | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); head -20 $f ;;
esac
done
