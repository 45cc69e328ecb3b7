#!/bin/bash
# Copy what an end-of-round GPU trip (tools/final_round.sh -> gpurun_out/final/) produced into profiles/<round>/ and refuse when it
# is older than the library it claims to describe: usage  tools/collect_round.sh r03
r=${1:?round, e.g. r03}; src=gpurun_out/final; dst=profiles/$r; mkdir -p $dst
lib=hirest_amd/lib/libhirest_hip.so
newest_src=$(ls -t hirest_amd/csrc/*.hip hirest_amd/csrc/*.h include/*.h | head -1)
if [ "$src/pmc_traffic.json" -ot "$newest_src" ]; then echo "STALE: $src/pmc_traffic.json is older than $newest_src: re-run tools/final_round.sh"; exit 1; fi
for f in bench_n1.json bench_under_rocprofv3.json rocprofv3_kernel_stats_bench.csv rocprofv3_kernel_trace_by_shape.csv pmc_traffic.json \
         pmc_bench_counters.txt pmc_table_bench.md secondary.json caption_word_timeline.txt caption_captions_per_s.txt pmc_caption_kernels.txt \
         fma_order_probe.txt wave_sum_probe.txt lane_path_probe.txt lm_head_probe.txt train_bench.txt rocprofv3_kernel_stats_train_step.csv \
         train_step_timeline.txt train_host_probe.txt gemm_f32_sweep.txt mfma_f32_peak_probe.txt caption_batch_breakdown.txt \
         x3_bench.json pmc_x3_kernels.txt pmc_table_x3.md no_profile_ab.txt caption_streams_sweep.txt grid_barrier_probe.txt c3_512_rank_blocks.json c3_n1.json \
         caption_captions_per_s_by_batch.txt lm_head_rows_ab.txt caption_word_timeline_b32_beam5.txt caption_word_timeline_b32_beam3.txt lds_rate_probe_run.txt gemm_f32_ring_ab_run.txt; do
  [ -f $src/$f ] && cp $src/$f $dst/$f
done
grep -E "passed|failed|error" $src/pytest_gpu.log | tail -3 > $dst/pytest_gpu_tail.txt
echo "collected into $dst:"; ls -la $dst | tail -n +2 | awk '{print "  " $NF, $5}'
