#!/bin/bash
# Step captioning on the GPU: token parity tests, captions/s, and the kernel timeline of one word.
out=gpurun_out/capt; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_joint.py -m gpu -q -x -k "caption or gemm_f32 or beam_tail or layernorm or attention or tile_maxima" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for i in 1 2 3; do for b in 5 3; do timeout 300 python tools/caption_profile.py $b 2>&1 | tail -1; done; done | tee $out/captions.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o x -- python $GRAFT_REPO_ROOT/tools/caption_profile.py 5 > /dev/null 2>&1 )
t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python tools/word_timeline.py "$t" > $out/word_timeline.txt 2>&1; cat $out/word_timeline.txt
rm -rf $out/prof
