#!/bin/bash
mkdir -p gpurun_out/run8; export TMPDIR=/tmp
O=gpurun_out/run8
for d in 0 1024 2048 3072 0; do
timeout 200 python tools/gemm_bench.py --variants 0 --shapes proj_stats fc2_stats qkv_fold fc1_fold --iters 20 --dbg $d 2>&1 | grep -v amdgpu.ids | tee -a $O/stagger.log
done
