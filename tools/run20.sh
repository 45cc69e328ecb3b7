#!/bin/bash
mkdir -p gpurun_out/run20
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -3
timeout 900 python tools/train_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run20/train_bench.txt
