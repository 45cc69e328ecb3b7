#!/bin/bash
# one word of the B = 32 beam-5 search, kernel by kernel, fp32 vs bf16x3 (LM head on split operands)
out=gpurun_out/capt6; mkdir -p $out; export TMPDIR=/tmp
for prec in fp32 bf16x3; do
( cd /tmp && HIREST_JOINT_PRECISION=$prec CAPTION_B=32 CAPTION_REPS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_$prec -o x -- python $GRAFT_REPO_ROOT/tools/caption_profile.py 5 > /dev/null 2>&1 )
t=$(find $out/prof_$prec -name "*kernel_trace.csv" | head -1)
echo "== $prec"; python tools/word_timeline.py "$t" > $out/word_timeline_b32_beam5_$prec.txt 2>&1; cat $out/word_timeline_b32_beam5_$prec.txt
rm -rf $out/prof_$prec
done
