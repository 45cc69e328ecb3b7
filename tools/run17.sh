#!/bin/bash
mkdir -p gpurun_out/run17
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lnfold.py -m gpu -q -x 2>&1 | tail -3
{
for dbg in 0 65536 131072 196608; do
  echo "== dbg $dbg (bit16 = old XCD split, bit17 = no slot rotation)"
  timeout 300 python tools/gemm_bench.py --variants 6 8 --iters 20 --shapes proj_plain proj_stats fc2_plain fc2_stats qkv --dbg $dbg 2>&1 | grep -v amdgpu.ids
done
echo "== zero-padded W rows"
timeout 300 python tools/gemm_bench.py --variants 6 8 --iters 20 --shapes fc2_plain fc2_stats proj_stats qkv --w-zero-pad 2>&1 | grep -v amdgpu.ids
echo "== zero operands (structure limit), new vs old schedule"
timeout 300 python tools/gemm_bench.py --variants 6 8 --iters 20 --shapes proj_plain fc2_plain qkv --a-scale 0 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/gemm_bench.py --variants 6 8 --iters 20 --shapes proj_plain fc2_plain qkv --a-scale 0 --dbg 196608 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/run17/schedule_ab.txt
