#!/bin/bash
# full GPU suite + smoke + default bench, then the operand-activity experiment (power cap vs kernel limit)
mkdir -p gpurun_out/run14
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/run14/pytest_gpu.log 2>&1; tail -4 gpurun_out/run14/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/run14/smoke.log 2>&1; tail -1 gpurun_out/run14/smoke.log
timeout 600 python bench.py > gpurun_out/run14/bench.json 2> gpurun_out/run14/bench.err; cut -c1-300 gpurun_out/run14/bench.json
for sc in "1 0.02" "0 0.02" "0 0" "1 0"; do set -- $sc
  echo "== A scale $1, W scale $2"
  timeout 300 python tools/gemm_bench.py --variants 6 8 --hipblaslt --iters 20 --shapes fc1_nogelu fc2_plain --a-scale $1 --w-scale $2 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/run14/operand_activity.txt
