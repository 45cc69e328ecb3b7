#!/bin/bash
# Round 5, step captioning with merged searches: parity tests, captions/s at B = 5 / 20 / 32, the LM-head kernels at 60 - 256 rows,
# and the kernel timeline of one word at B = 32.
out=gpurun_out/capt5; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_joint.py -m gpu -q -x -k "${CAPT_K:-caption or gemm_f32 or beam_tail or tile_maxima or lm_head or moment_model}" > $out/pytest.log 2>&1; tail -5 $out/pytest.log
timeout 300 python tools/lm_head_rows_ab.py 2>&1 | tee $out/lm_head_rows_ab.txt
for B in 5 20 32; do for b in 5 3; do CAPTION_B=$B timeout 300 python tools/caption_profile.py $b 2>&1 | tail -1; done; done | tee $out/captions.txt
for b in 5 3; do
( cd /tmp && CAPTION_B=32 CAPTION_REPS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof$b -o x -- python $GRAFT_REPO_ROOT/tools/caption_profile.py $b > /dev/null 2>&1 )
t=$(find $out/prof$b -name "*kernel_trace.csv" | head -1)
python tools/word_timeline.py "$t" > $out/word_timeline_b32_beam$b.txt 2>&1; cat $out/word_timeline_b32_beam$b.txt
rm -rf $out/prof$b
done
