#!/bin/bash
mkdir -p gpurun_out/run5; export TMPDIR=/tmp
O=gpurun_out/run5
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lnfold.py -q -x -k "w4h" > $O/pytest_w4h.log 2>&1; tail -3 $O/pytest_w4h.log
timeout 250 python tools/gemm_stress.py --variant 17 --cases 40 --repeats 3 > $O/stress_w4h.log 2>&1; tail -2 $O/stress_w4h.log
timeout 200 python tools/gemm_stress.py --variant 17 --full --cases 8 --repeats 3 > $O/stress_w4h_full.log 2>&1; tail -2 $O/stress_w4h_full.log
timeout 300 python tools/gemm_bench.py --variants 6 8 17 --shapes qkv fc1_nogelu proj_plain fc2_plain proj proj_stats fc2 fc2_stats --iters 10 > $O/gemm.log 2>&1; cat $O/gemm.log
