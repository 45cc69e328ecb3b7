#!/bin/bash
# GPU trip for the d2 kernel (two 256x128 workgroups per CU): parity, race screen, then timing against the default dispatch.
mkdir -p gpurun_out/d2
O=gpurun_out/d2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lnfold.py -q -x -k "d2" 2>&1 | tail -8 > $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python tools/gemm_stress.py --variant 18 --cases 30 --repeats 3 2>&1 | tail -4 > $O/stress.log; tail -2 $O/stress.log
timeout 600 python tools/gemm_bench.py --variants 0 18 20 19 --iters 10 --shapes qkv proj_plain proj_stats fc2_plain 2>&1 | tee -a $O/bench.log
timeout 600 python tools/gemm_bench.py --variants 0 18 --iters 10 --shapes qkv_fold fc1_fold fc2_stats 2>&1 | tee -a $O/bench.log
