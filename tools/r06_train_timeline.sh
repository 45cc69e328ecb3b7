#!/bin/bash
# Round 6: kernel timeline of one training step (T = 300, retrieval, fused AdamW so the optimizer's share is small) with the encoder blocks
# issued from C, fp32 products and bf16x3 products.   gpurun -- 'bash tools/r06_train_timeline.sh'
out=gpurun_out/train_tl; mkdir -p $out; export TMPDIR=/tmp
for g in fp32 bf16x3; do
  ( cd /tmp && HIREST_TRAIN_GEMM=$g timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_$g -o x -- python $GRAFT_REPO_ROOT/tools/train_bench.py --frames 300 --tasks moment_retrieval --reps 8 > /dev/null 2>&1 )
  t=$(find $out/prof_$g -name "*kernel_trace.csv" | head -1)
  python tools/step_timeline.py "$t" --list > $out/step_timeline_$g.txt 2>&1
  grep "^#\| x " $out/step_timeline_$g.txt | head -40
  rm -rf $out/prof_$g
done
