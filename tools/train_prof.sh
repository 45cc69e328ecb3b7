#!/bin/bash
# rocprofv3 kernel stats of the joint-model training step (moment retrieval, B = 5, T = 300)
mkdir -p gpurun_out/train
O=gpurun_out/train
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o t -- python $GRAFT_REPO_ROOT/tools/train_bench.py --frames 300 --reps 10 --tasks moment_retrieval > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/train_kernel_stats.csv; head -34 $f | cut -c1-150
grep -v amdgpu.ids $O/prof.log | tail -3
rm -rf $O/prof
