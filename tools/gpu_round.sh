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
gemm)
  timeout 600 python tools/gemm_bench.py --shapes qkv fc2 2>&1 | tee gpurun_out/gemm_bench.log
  timeout 600 python tools/gemm_bench.py --shapes qkv fc2 --alias 2>&1 | tee -a gpurun_out/gemm_bench.log ;;
pmc)
  cd /tmp
  rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -o p -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py --variants 2 --iters 3 --shapes qkv fc2 > $GRAFT_REPO_ROOT/gpurun_out/pmc1.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o p -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py --variants 2 --iters 3 --shapes qkv fc2 > $GRAFT_REPO_ROOT/gpurun_out/pmc2.log 2>&1
  cd $GRAFT_REPO_ROOT
  python - <<'PY'
import csv, glob, collections
for d in ("pmc1","pmc2"):
    fs = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    print(d, fs)
    for f in fs:
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name","")[:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, v in agg.items():
            if "gemm" not in k: continue
            print(k)
            for c, val in v.items(): print(f"    {c:28s} {val / max(cnt[(k,c)],1):.4g} (per dispatch, n={cnt[(k,c)]})")
PY
  ;;
bench)
  timeout 900 python bench.py > gpurun_out/bench.log 2>&1; tail -c 2500 gpurun_out/bench.log ;;
prof)
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; cd $GRAFT_REPO_ROOT
  ls -R gpurun_out/prof | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); head -20 $f ;;
esac
done
