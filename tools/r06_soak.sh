#!/bin/bash
# Soak (round 6): the one test that failed once in a full-suite run (rank blocks != single sweep), repeated in fresh processes.
N=${1:-12}
mkdir -p gpurun_out/soak
pass=0; fail=0
for i in $(seq 1 $N); do
  if timeout 600 python -m pytest tests/test_run_corpus.py -q -x -k "c3_eight_rank_blocks" > gpurun_out/soak/c3_$i.log 2>&1; then pass=$((pass+1)); else fail=$((fail+1)); tail -5 gpurun_out/soak/c3_$i.log; fi
done
echo "c3 rank-block test: $pass passed, $fail failed of $N" | tee gpurun_out/soak/summary.txt
