#!/bin/bash
mkdir -p gpurun_out/run18
timeout 900 python tools/train_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run18/train_bench.txt
