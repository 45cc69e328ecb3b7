#!/bin/bash
# Round 6: the training step with its encoder blocks issued from C (csrc/train_block.hip) against the per-kernel Python calls, and with the
# blocks' forward / dX products on split operands (HIREST_TRAIN_GEMM=bf16x3), same box.
mkdir -p gpurun_out/train_c
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_joint_x3.py -x -q -k "not encoder_x3 and not gemm_x3_t128" 2>&1 | tail -5 | tee gpurun_out/train_c/pytest.txt
for v in "0 fp32" "1 fp32" "1 bf16x3"; do
  set -- $v
  echo "== HIREST_TRAIN_C_BLOCKS=$1 HIREST_TRAIN_GEMM=$2" | tee -a gpurun_out/train_c/bench.txt
  HIREST_TRAIN_C_BLOCKS=$1 HIREST_TRAIN_GEMM=$2 timeout 600 python tools/train_bench.py --frames 120 300 --fused --tasks moment_retrieval moment_segmentation 2>&1 | grep "T=" | tee -a gpurun_out/train_c/bench.txt
done
