#!/bin/bash
mkdir -p gpurun_out/run19
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/run19/prof -- python $GRAFT_REPO_ROOT/tools/train_bench.py --frames 300 --reps 20 > $GRAFT_REPO_ROOT/gpurun_out/run19/log.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/run19/prof -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-200
