#!/bin/bash
for i in 1 2; do timeout 300 python tools/gemm_bench.py --variants 6 --iters 30 --shapes fc1 fc1_fold fc1_nogelu 2>&1 | grep -v amdgpu.ids; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gemm or gelu" 2>&1 | tail -2
