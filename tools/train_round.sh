#!/bin/bash
mkdir -p gpurun_out/train
O=gpurun_out/train
timeout 900 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -6 > $O/pytest.log; tail -6 $O/pytest.log
timeout 600 python tools/train_bench.py --fused --frames 120 300 571 2>&1 | grep -v amdgpu.ids | tee $O/train_bench.txt
