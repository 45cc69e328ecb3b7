#!/bin/bash
# GPU trip: w4 kernel correctness + timing against p256 / pp256 / hipBLASLt, then the bench line.  Every step has its own timeout
# and writes its log as it goes.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/run2; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lnfold.py -q -x -k "w4 or lnfold or consumer or producer or guard or reverse" > $O/pytest_w4.log 2>&1; tail -5 $O/pytest_w4.log
timeout 300 python tools/gemm_bench.py --variants 6 8 9 --shapes qkv_fold fc1_fold proj_stats fc2_stats --iters 10 > $O/gemm_fold.log 2>&1; cat $O/gemm_fold.log
timeout 300 python tools/gemm_bench.py --variants 6 9 10 11 13 15 --shapes qkv fc1_nogelu proj_plain fc2_plain --iters 10 --hipblaslt > $O/gemm_plain.log 2>&1; cat $O/gemm_plain.log
timeout 200 python tools/gemm_stress.py --variant 9 --cases 30 --repeats 3 > $O/stress_w4.log 2>&1; tail -3 $O/stress_w4.log
timeout 300 python bench.py --steps 5 --warmup 2 --gemm-kernel 9 --no-cpu-baseline --no-matched-recall > $O/bench_w4.log 2>$O/bench_w4.err; python - <<'PY'
import json
for f in ("gpurun_out/run2/bench_w4.log",):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, "frames/s %.1f"%d["value"])
        for e in d["roofline"]["breakdown"][:8]: print("   ",e["kernel"],e["tag"],e["dims"],"avg_ms %.3f"%e["avg_ms"],"share %.3f"%e["share"],"TF %.0f"%e.get("tflops",0))
    except Exception as ex: print(f,"failed",ex); print(open(f.replace('.log','.err')).read()[-1500:])
PY
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench_default.log 2>$O/bench_default.err; tail -c 3000 $O/bench_default.log; tail -5 $O/bench_default.err
