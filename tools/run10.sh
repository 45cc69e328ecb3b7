#!/bin/bash
mkdir -p gpurun_out/run10
for d in 0 4096 8192 16384 32768 12288 61440 0; do
timeout 200 python tools/gemm_bench.py --variants 0 --shapes proj_stats fc2_stats --iters 20 --dbg $d 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/run10/epi.log
done
