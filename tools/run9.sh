#!/bin/bash
mkdir -p gpurun_out/run9
timeout 600 python -m pytest tests/test_gpu_train.py -q -x -s > gpurun_out/run9/pytest_train.log 2>&1; tail -30 gpurun_out/run9/pytest_train.log | cut -c1-220
