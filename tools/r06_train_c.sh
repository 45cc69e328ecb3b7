#!/bin/bash
# Round 6: the training step with its encoder blocks issued from C (csrc/train_block.hip) against the per-kernel Python calls, same box.
mkdir -p gpurun_out/train_c
timeout 900 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -5 | tee gpurun_out/train_c/pytest.txt
for v in 0 1; do
  echo "== HIREST_TRAIN_C_BLOCKS=$v" | tee -a gpurun_out/train_c/bench.txt
  HIREST_TRAIN_C_BLOCKS=$v timeout 600 python tools/train_bench.py --frames 120 300 --fused --tasks moment_retrieval moment_segmentation 2>&1 | grep "T=" | tee -a gpurun_out/train_c/bench.txt
done
