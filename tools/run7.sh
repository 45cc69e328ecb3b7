#!/bin/bash
mkdir -p gpurun_out/run7; export TMPDIR=/tmp
O=gpurun_out/run7
for d in 0 1024 2048 3072 0 2048; do
timeout 200 python tools/gemm_bench.py --variants 0 --shapes proj_stats fc2_stats --iters 20 --dbg $d 2>&1 | grep -v amdgpu.ids | tee -a $O/rd.log
done
timeout 200 python -m pytest tests/test_gpu_lnfold.py -q -x -k "producer or reverse" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
