#!/bin/bash
# Round 6: margin-guarded re-rank — GPU parity test, then the re-encoded fraction / effective frames/s table (512 x 32 and 4096 x 32 corpora)
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_rank_exact.py -m gpu -q -s 2>&1 | grep -v amdgpu.ids | tail -6
for k in 1 5 10; do
  echo "== 512 x 32, k = $k"
  timeout 900 python tools/c3_run.py --videos 512 --frames 32 --rank-exact-k $k --verify-rank-exact --out gpurun_out/rank_exact_512_k$k.json 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(r['rank_exact']))"
done
echo "== 4096 x 32, k = 1 (verified against the whole corpus at bf16x3)"
timeout 1500 python tools/c3_run.py --videos 4096 --frames 32 --rank-exact-k 1 --verify-rank-exact --out gpurun_out/rank_exact_4096_k1.json 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(r['rank_exact']))"
echo "== 4096 x 32, k = 10"
timeout 1500 python tools/c3_run.py --videos 4096 --frames 32 --rank-exact-k 10 --out gpurun_out/rank_exact_4096_k10.json 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(r['rank_exact']))"
} 2>&1 | tee gpurun_out/r06_rank_exact.txt
