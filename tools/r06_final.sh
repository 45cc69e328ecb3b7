#!/bin/bash
# Round 6 end-of-round GPU trip: full GPU suite, smoke, the default bench line, rocprofv3 kernel stats of the bench, memory-side traffic, SQ counters,
# race screens, configs[2] runs.  Outputs under gpurun_out/final/ (tools/collect_round.sh r06 copies what is to be judged into profiles/r06/).
out=gpurun_out/final; mkdir -p $out; export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q ) > $out/pytest_gpu.log 2>&1; tail -6 $out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -1 $out/smoke.log
( time timeout 1200 python bench.py > $out/bench_n1.json 2> $out/bench.err ) 2>&1 | grep real; cut -c1-300 $out/bench_n1.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o r -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-matched-recall --no-secondary > $GRAFT_REPO_ROOT/$out/bench_under_rocprofv3.json 2> $GRAFT_REPO_ROOT/$out/prof.err )
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp "$f" $out/rocprofv3_kernel_stats_bench.csv; head -8 $out/rocprofv3_kernel_stats_bench.csv | cut -c1-150
t=$(find $out/prof -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/trace_summary.py "$t" > $out/rocprofv3_kernel_trace_by_shape.csv 2>/dev/null
rm -rf $out/prof
bash tools/pmc_traffic.sh $out/pmc_traffic.json > $out/pmc_traffic.log 2>&1; head -c 600 $out/pmc_traffic.json; echo
bash tools/pmc_sq.sh $out/pmcsq "gemm_|attention_kernel" -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-matched-recall --no-secondary > $out/pmc_bench_counters.txt 2>&1
python tools/pmc_summary.py $out/pmc_bench_counters.txt > $out/pmc_table_bench.md; rm -rf $out/pmcsq; cat $out/pmc_table_bench.md | cut -c1-200
timeout 600 python tools/tower_stress.py --repeats 20 2>&1 | tail -3 | tee $out/tower_stress.txt
REPS=6 timeout 600 python tools/r06_race.py 2>&1 | grep -v amdgpu.ids | tee $out/race_screen.txt
timeout 300 python tools/c3_run.py --videos 512 --rank-blocks 8 --out $out/c3_512_rank_blocks.json > /dev/null 2>&1; python -c "import json; d=json.load(open('$out/c3_512_rank_blocks.json')); print({k: d[k] for k in ('frames_per_s_incl_input_generation','equals_committed_1rank_digest','pooled_rows_bit_identical','top10_identical')})"
timeout 400 python tools/c3_run.py --out $out/c3_n1.json > /dev/null 2>&1; python -c "import json; d=json.load(open('$out/c3_n1.json')); print({k: d[k] for k in ('frames_per_s_incl_input_generation','equals_committed_1rank_digest','seconds_incl_input_generation')})"
timeout 300 python tools/x3_bench.py > $out/x3_bench.json 2> /dev/null; head -c 300 $out/x3_bench.json; echo
date +%s > $out/collected_at
