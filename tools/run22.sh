#!/bin/bash
mkdir -p gpurun_out/run22
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/run22/prof -- python $GRAFT_REPO_ROOT/tools/caption_profile.py > $GRAFT_REPO_ROOT/gpurun_out/run22/log.txt 2>&1
cd $GRAFT_REPO_ROOT
grep captioning gpurun_out/run22/log.txt
f=$(find gpurun_out/run22/prof -name "*kernel_stats.csv" | head -1)
head -16 "$f" | cut -c1-170
