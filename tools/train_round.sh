#!/bin/bash
# Joint-model training step on the GPU: gradient parity tests, ms per step, and the kernel timeline of one step (T = 300, retrieval).
out=gpurun_out/train; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -x > $out/pytest.log 2>&1; tail -2 $out/pytest.log
timeout 600 python tools/train_bench.py --frames 120 300 571 --fused 2>&1 | grep -v amdgpu.ids | tee $out/train_bench.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o x -- python $GRAFT_REPO_ROOT/tools/train_bench.py --frames 300 --tasks moment_retrieval --reps 8 > /dev/null 2>&1 )
t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py "$t" ${1:+--list} > $out/step_timeline.txt 2>&1; cat $out/step_timeline.txt
rm -rf $out/prof
