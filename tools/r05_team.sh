#!/bin/bash
# Round 5: the team walk of the persistent GEMM (hirest_gemm_debug_mode bit 18) — bit-equality screen and A/B timing on the proj / fc2 shapes
out=gpurun_out/team; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python tools/gemm_stress.py --variant 9 --dbg 262144 --cases 30 2>&1 | tail -4 | tee $out/stress.txt
timeout 300 python tools/gemm_stress.py --variant 9 --dbg 262144 --full --cases 4 --repeats 2 2>&1 | tail -2 | tee -a $out/stress.txt
for i in 1 2; do
for d in 0 262144; do timeout 300 python tools/gemm_bench.py --variants 0 --dbg $d --shapes proj_stats2 fc2_stats2 fc2 proj --iters 20 2>&1 | grep -v amdgpu; done
done | tee $out/ab.txt
