#!/bin/bash
# GPU trip: SQ counters of the attention kernel, and of d2 vs p256 (pmc_sq.sh runs its command from /tmp: absolute paths).
mkdir -p gpurun_out/attn
O=gpurun_out/attn
R=$GRAFT_REPO_ROOT
bash tools/pmc_sq.sh gpurun_out/attn/pmc "attention_kernel_v3" -- python $R/tools/attn_bench.py --variants 3 --iters 3 > $O/pmc_attn.txt 2>&1; head -40 $O/pmc_attn.txt
bash tools/pmc_sq.sh gpurun_out/attn/pmc_d2 "gemm_d2|gemm_p256" -- python $R/tools/gemm_bench.py --variants 0 18 20 19 --iters 3 --warmup 3 --shapes proj_stats proj_plain qkv > $O/pmc_d2.txt 2>&1; head -150 $O/pmc_d2.txt
rm -rf gpurun_out/attn/pmc gpurun_out/attn/pmc_d2
