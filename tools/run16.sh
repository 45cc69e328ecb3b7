#!/bin/bash
# L2 set-aliasing experiment: does padding the operands' row strides (12288 B = 96 lines for K = 6144) change the GEMM time?
mkdir -p gpurun_out/run16
for pads in "0 0" "64 0" "0 64" "64 64" "32 32" "8 8"; do set -- $pads
  for sc in 1 0; do
    echo "== lda+$1 ldw+$2 operand scale $sc"
    timeout 300 python tools/gemm_bench.py --variants 6 8 --hipblaslt --iters 20 --shapes fc2_plain fc1_nogelu --lda-pad $1 --ldw-pad $2 --a-scale $sc 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/run16/stride_pad.txt
