#!/bin/bash
# The core of tools/final_round.sh (smoke, default bench line, rocprofv3 kernel stats, memory-side traffic, SQ counters): a mid-round checkpoint
# Outputs under gpurun_out/final/ (copy what should be judged into profiles/rNN/).
out=gpurun_out/final; mkdir -p $out; export TMPDIR=/tmp
echo "(GPU suite run separately)" > $out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 900 python bench.py > $out/bench_n1.json 2> $out/bench.err; cut -c1-260 $out/bench_n1.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o r -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-matched-recall --no-secondary > $GRAFT_REPO_ROOT/$out/bench_under_rocprofv3.json 2> $GRAFT_REPO_ROOT/$out/prof.err )
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp "$f" $out/rocprofv3_kernel_stats_bench.csv; head -8 $out/rocprofv3_kernel_stats_bench.csv | cut -c1-150
t=$(find $out/prof -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/trace_summary.py "$t" > $out/rocprofv3_kernel_trace_by_shape.csv 2>/dev/null
rm -rf $out/prof
bash tools/pmc_traffic.sh $out/pmc_traffic.json > $out/pmc_traffic.log 2>&1; head -c 600 $out/pmc_traffic.json
# SQ / TCC counters of the kernels the bench runs (table for profiles/rNN/pmc_summary.md)
