#!/bin/bash
mkdir -p gpurun_out/run12
timeout 300 python tools/attn_bench.py --variants 3 --iters 20 --dbg 0 1 2 4 8 7 15 0 2>&1 | grep -v amdgpu | tee gpurun_out/run12/attn_dbg.log
