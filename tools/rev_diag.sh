#!/bin/bash
# FETCH_SIZE and time of the production GEMMs walking their tile list forwards vs backwards (HIREST_GEMM_REVERSE);
# findings: DESIGN.md 4.1b
for sh in fc2_stats proj_stats qkv_fold fc1_fold; do for r in "" "--reverse"; do
  echo "== $sh $r"
  bash tools/pmc_fetch.sh gemm_p -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py --variants 0 --iters 3 --warmup 2 --shapes $sh $r | tail -1 | cut -c1-150
done; done
rm -rf gpurun_out/pmcf
for sh in fc2_stats proj_stats qkv_fold fc1_fold; do for r in "" "--reverse" "" "--reverse"; do timeout 200 python tools/gemm_bench.py --variants 0 --iters 20 --shapes $sh $r 2>&1 | grep TFLOP | sed "s/^/$r /"; done; done
