#!/bin/bash
# PMC comparison on the fc1 / fc2 plain shapes: p256 (6), pp256 (8), w4 (9), w4 without LDS-DMA (10), hipBLASLt
mkdir -p gpurun_out/run3; export TMPDIR=/tmp
O=gpurun_out/run3
for sh in fc1_nogelu fc2_plain; do
  bash tools/pmc_sq.sh $O/pmc_$sh "gemm_p256|gemm_pp256|gemm_w4|Cijk" -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py --variants 6 8 9 10 --shapes $sh --iters 3 --warmup 5 --hipblaslt > $O/pmc_$sh.txt 2>&1
  cat $O/pmc_$sh.txt | cut -c1-200
done
