#!/bin/bash
mkdir -p gpurun_out/run4; export TMPDIR=/tmp
O=gpurun_out/run4
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lnfold.py -q -x -k "w4h" > $O/pytest_w4h.log 2>&1; tail -3 $O/pytest_w4h.log
timeout 200 python tools/gemm_stress.py --variant 17 --cases 30 --repeats 3 > $O/stress_w4h.log 2>&1; tail -2 $O/stress_w4h.log
timeout 300 python tools/gemm_bench.py --variants 6 8 9 17 --shapes qkv fc1_nogelu proj_plain fc2_plain --iters 10 > $O/gemm_plain.log 2>&1; cat $O/gemm_plain.log
timeout 300 python tools/gemm_bench.py --variants 6 8 17 --shapes qkv_fold fc1_fold proj_stats fc2_stats --iters 10 > $O/gemm_fold.log 2>&1; cat $O/gemm_fold.log
timeout 300 python bench.py --steps 5 --warmup 2 --gemm-kernel 17 --no-cpu-baseline --no-matched-recall > $O/bench_w4h.log 2>$O/bench_w4h.err; python - <<'PY'
import json
f="gpurun_out/run4/bench_w4h.log"
try:
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, "frames/s %.1f"%d["value"])
    for e in d["roofline"]["breakdown"][:6]: print("   ",e["kernel"],e["tag"],e["dims"],"avg_ms %.3f"%e["avg_ms"],"TF %.0f"%e.get("tflops",0))
except Exception as ex: print(f,"failed",ex); print(open(f.replace('.log','.err')).read()[-1500:])
PY
for sh in fc1_nogelu fc2_plain; do
  bash tools/pmc_sq.sh $O/pmc_$sh "gemm_w4" -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py --variants 17 --shapes $sh --iters 3 --warmup 5 > $O/pmc_$sh.txt 2>&1
done
