#!/bin/bash
# Round 6: host enqueue time per phase of the training step (tools/train_host_probe.py), per-kernel Python calls vs C blocks vs C blocks at bf16x3
mkdir -p gpurun_out/train_c
for v in "0 fp32" "1 fp32" "1 bf16x3"; do
  set -- $v
  echo "== HIREST_TRAIN_C_BLOCKS=$1 HIREST_TRAIN_GEMM=$2"
  HIREST_TRAIN_C_BLOCKS=$1 HIREST_TRAIN_GEMM=$2 timeout 300 python tools/train_host_probe.py 2>&1 | grep -v amdgpu | head -12
done 2>&1 | tee gpurun_out/train_c/host_probe.txt
