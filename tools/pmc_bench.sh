#!/bin/bash
# SQ / TCC counters of the kernels bench.py runs (four separate --pmc passes, kernel trace only): raw per-kernel averages + the table
out=gpurun_out/pmcb; rm -rf $out; mkdir -p $out
bash tools/pmc_sq.sh $out "gemm_|attention_kernel" -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-matched-recall > $out/pmc_bench_counters.txt 2>&1
python tools/pmc_summary.py $out/pmc_bench_counters.txt > $out/table1.md; cat $out/table1.md
rm -rf $out/p0 $out/p1 $out/p2 $out/p3
