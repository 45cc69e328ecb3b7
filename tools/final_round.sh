#!/bin/bash
# End-of-round GPU trip: full GPU suite, smoke, the default bench line, rocprofv3 kernel stats of the bench, HBM-side traffic.
# Outputs under gpurun_out/final/ (copy what should be judged into profiles/rNN/).
out=gpurun_out/final; mkdir -p $out; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 900 python bench.py > $out/bench_n1.json 2> $out/bench.err; cut -c1-260 $out/bench_n1.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o r -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-matched-recall --no-secondary > $GRAFT_REPO_ROOT/$out/bench_under_rocprofv3.json 2> $GRAFT_REPO_ROOT/$out/prof.err )
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp "$f" $out/rocprofv3_kernel_stats_bench.csv; head -8 $out/rocprofv3_kernel_stats_bench.csv | cut -c1-150
t=$(find $out/prof -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/trace_summary.py "$t" > $out/rocprofv3_kernel_trace_by_shape.csv 2>/dev/null
rm -rf $out/prof
bash tools/pmc_traffic.sh $out/pmc_traffic.json > $out/pmc_traffic.log 2>&1; head -c 600 $out/pmc_traffic.json
# SQ / TCC counters of the kernels the bench runs (table for profiles/rNN/pmc_summary.md)
bash tools/pmc_sq.sh $out/pmcsq "gemm_|attention_kernel" -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-matched-recall --no-secondary > $out/pmc_bench_counters.txt 2>&1
python tools/pmc_summary.py $out/pmc_bench_counters.txt > $out/pmc_table_bench.md; rm -rf $out/pmcsq; cat $out/pmc_table_bench.md | cut -c1-200
python tools/secondary_bench.py > $out/secondary.json 2> $out/secondary.err; head -c 400 $out/secondary.json
# step captioning: captions/s (3 runs), kernel timeline of one word, SQ counters of its kernels; the two probes its kernels rest on
bash tools/caption_round.sh > $out/caption_round.log 2>&1; cp gpurun_out/capt/word_timeline.txt $out/caption_word_timeline.txt; cp gpurun_out/capt/captions.txt $out/caption_captions_per_s.txt
CAPTION_REPS=1 bash tools/pmc_sq.sh $out/pmccap "m16|attention_f32_decode|tail_" -- python $GRAFT_REPO_ROOT/tools/caption_profile.py 5 > $out/pmc_caption_kernels.txt 2>&1; rm -rf $out/pmccap
for pr in fma_order_probe wave_sum_probe lane_path_probe; do hipcc --offload-arch=gfx950 -O2 -w -o /tmp/$pr tools/probes/$pr.hip && timeout 60 /tmp/$pr > $out/$pr.txt 2>&1; done
timeout 120 python tools/lm_head_probe.py > $out/lm_head_probe.txt 2>&1
# training step: timings (reference optimizer call and fused=True), kernel stats of the retrieval step
bash tools/train_round.sh > $out/train_round.log 2>&1; cp gpurun_out/train/train_bench.txt $out/train_bench.txt; cp gpurun_out/train/step_timeline.txt $out/train_step_timeline.txt
timeout 300 python tools/train_host_probe.py 2>&1 | grep -v amdgpu.ids | head -60 > $out/train_host_probe.txt
timeout 300 python tools/gemm_f32_sweep.py 2>&1 | grep -v amdgpu.ids > $out/gemm_f32_sweep.txt
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_f32_peak_probe tools/probes/mfma_f32_peak_probe.hip && timeout 120 /tmp/mfma_f32_peak_probe > $out/mfma_f32_peak_probe.txt 2>&1
for b in 5 3; do timeout 300 python tools/caption_batch_breakdown.py $b 2>&1 | tail -1; done > $out/caption_batch_breakdown.txt
bash tools/train_prof.sh > $out/train_prof.log 2>&1; cp gpurun_out/train/train_kernel_stats.csv $out/rocprofv3_kernel_stats_train_step.csv
# round 4: the bf16x3 tower (per-kernel breakdown + SQ counters), the profile-instrumentation A/B, captioning with batches in flight, the
# grid-barrier probe, configs[2] at 512 x 32 with its 8-rank-block invariance
timeout 300 python tools/x3_bench.py > $out/x3_bench.json 2> /dev/null
bash tools/pmc_sq.sh $out/pmcx3 "gemm_pp256x3|attention_x3|attention_f32|split2" -- python $GRAFT_REPO_ROOT/tools/x3_bench.py --frames 512 --steps 1 > $out/pmc_x3_kernels.txt 2>&1
python tools/pmc_summary.py $out/pmc_x3_kernels.txt > $out/pmc_table_x3.md; rm -rf $out/pmcx3
for i in 1 2; do for f in "" "--no-profile"; do timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-matched-recall --no-secondary $f 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench.py $f:', round(d['value'], 1), 'frames/s', round(d['ms_per_step'], 2), 'ms per step')"; done; done > $out/no_profile_ab.txt 2>&1
timeout 400 python tools/caption_streams_sweep.py 2>&1 | grep -v amdgpu.ids > $out/caption_streams_sweep.txt
hipcc --offload-arch=gfx950 -O2 -w -o /tmp/gbp tools/probes/grid_barrier_probe.hip && timeout 120 /tmp/gbp > $out/grid_barrier_probe.txt 2>&1
timeout 300 python tools/c3_run.py --videos 512 --rank-blocks 8 --out $out/c3_512_rank_blocks.json > /dev/null 2>&1
timeout 400 python tools/c3_run.py --out $out/c3_n1.json > /dev/null 2>&1
# round 5: merged / B = 32 captioning (captions/s, word timelines, the LM-head kernels by row count), the LDS rate probe, the fp32 GEMM's operand paths
CAPT_K="nothing_selected" bash tools/r05_caption.sh > $out/r05_caption.log 2>&1
cp gpurun_out/capt5/captions.txt $out/caption_captions_per_s_by_batch.txt; cp gpurun_out/capt5/lm_head_rows_ab.txt $out/lm_head_rows_ab.txt
cp gpurun_out/capt5/word_timeline_b32_beam5.txt $out/caption_word_timeline_b32_beam5.txt; cp gpurun_out/capt5/word_timeline_b32_beam3.txt $out/caption_word_timeline_b32_beam3.txt
hipcc --offload-arch=gfx950 -O2 -w -o /tmp/lds_rate_probe tools/probes/lds_rate_probe.hip && timeout 60 /tmp/lds_rate_probe > $out/lds_rate_probe_run.txt 2>&1
timeout 300 python tools/gemm_f32_ring_ab.py 2>&1 | grep -v amdgpu.ids > $out/gemm_f32_ring_ab_run.txt
date +%s > $out/collected_at
