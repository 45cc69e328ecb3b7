#!/bin/bash
# Round-2 profile set for profiles/r02 (final kernels) + re-validation of w4 schedule H after the barrier-count fix
mkdir -p gpurun_out/run6; export TMPDIR=/tmp
O=gpurun_out/run6
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lnfold.py -q -x -k "w4" > $O/pytest_w4.log 2>&1; tail -3 $O/pytest_w4.log
timeout 250 python tools/gemm_stress.py --variant 17 --cases 40 --repeats 3 > $O/stress_w4h.log 2>&1; tail -2 $O/stress_w4h.log
# 1. bench line
timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench_n1.json 2>$O/bench_n1.err; tail -c 600 $O/bench_n1.json
# 2. kernel trace + stats of the same command
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-matched-recall > $GRAFT_REPO_ROOT/$O/bench_under_rocprofv3.json 2>$GRAFT_REPO_ROOT/$O/prof.err )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/rocprofv3_kernel_stats_bench.csv; head -12 $f | cut -c1-200
t=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/trace_summary.py $t > $O/rocprofv3_kernel_trace_by_shape.csv 2>&1; head -12 $O/rocprofv3_kernel_trace_by_shape.csv | cut -c1-200
rm -rf $O/prof
# 3. SQ counters on the final kernels (separate passes, kernel trace only)
bash tools/pmc_sq.sh $O/pmc_sq "gemm_p256|gemm_pp256|attention_kernel_v3" -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-matched-recall > $O/pmc_bench_sq.txt 2>&1; head -60 $O/pmc_bench_sq.txt | cut -c1-150
rm -rf $O/pmc_sq/p*/ 
# 4. memory-side traffic
bash tools/pmc_traffic.sh $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1; tail -15 $O/pmc_traffic.txt | cut -c1-200
rm -rf gpurun_out/pmct
