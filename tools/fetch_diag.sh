#!/bin/bash
# FETCH_SIZE (memory-side L2 requests) per GEMM launch for column counts around the production shapes: does the
# tile schedule reach the a x b patch ideal?  Findings: DESIGN.md 4.1 item 6.
for spec in "4096 1408 0" "4224 1408 0" "4352 1408 0" "1280 6144 3" "1408 6144 3" "1536 6144 3" "1280 1408 3" "1408 1408 3" "1536 1408 3" "6144 1408 1"; do
  set -- $spec
  echo "== N=$1 K=$2 epi=$3"
  bash tools/pmc_fetch.sh gemm_p -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py --variants 0 --iters 3 --warmup 2 --shapes --nk $1 $2 $3 | tail -2
  grep TFLOP gpurun_out/pmcf.log | tail -1
done
rm -rf gpurun_out/pmcf
