#!/bin/bash
# HISTORICAL (round 6): this script measured the build that had the ragged column's own kernel (gemm_pq256r; removed again — it lost, see
# profiles/r06/ragged_column_kernel_ab.txt and DESIGN 4.1i).  On the current tree debug bit 20 means nothing and both arms run the same code.
# Round 6: the ragged column's own kernel (gemm_pq256r) — parity, race screens, then A/B against one launch (hirest_gemm_debug_mode bit 20) on the same box
mkdir -p gpurun_out
{
python tools/r06_dbg_epi.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_lnfold.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -3
timeout 600 python tools/gemm_stress.py --variant 9 --cases 40 --repeats 3 2>&1 | grep -v amdgpu.ids | tail -4
REPS=5 timeout 600 python tools/r06_race.py 2>&1 | grep -v amdgpu.ids
for round in 1 2; do for dbg in 1048576 0; do
  echo "== gemm_bench, debug mode $dbg (1048576 = one launch, no separate ragged column)"
  timeout 300 python tools/gemm_bench.py --variants 0 --iters 20 --dbg $dbg --shapes proj_stats2 fc2_stats2 qkv_fold 2>&1 | grep -v amdgpu.ids | tail -3
done; done
for round in 1 2; do for dbg in 1048576 0; do
  echo "== bench, debug mode $dbg"
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-matched-recall --no-secondary --gemm-dbg $dbg > gpurun_out/bb.log 2>&1
  python - <<PY
import json
x=json.loads(open("gpurun_out/bb.log").read().strip().splitlines()[-1])
print("frames/s %.0f"%x["value"], sorted([(e["tag"],e["dims"][1],e["dims"][2],round(e["avg_ms"],3)) for e in x["roofline"]["breakdown"][:6]]))
PY
done; done
timeout 600 python tools/tower_stress.py --repeats 10 2>&1 | tail -2
timeout 600 python -m pytest tests/test_run_corpus.py -m gpu -q -k "eight_rank" 2>&1 | tail -2
} 2>&1 | tee gpurun_out/r06_rag.txt
