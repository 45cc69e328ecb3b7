#!/bin/bash
mkdir -p gpurun_out/run11
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "test_attention" > gpurun_out/run11/pytest.log 2>&1; tail -3 gpurun_out/run11/pytest.log
timeout 200 python tools/attn_bench.py --variants 3 4 3 4 --iters 20 2>&1 | grep -v amdgpu | tee gpurun_out/run11/attn.log
