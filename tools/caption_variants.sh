#!/bin/bash
# captions/s for tuning variants of the decode GEMM (environment switches read once per process)
for v in "" "HIREST_M16_MID=1" "HIREST_M16_MID=2" "HIREST_M16_MID=3" "HIREST_M16_LM=1" "HIREST_M16_LM=2" "HIREST_M16_LM=3" "HIREST_M16_LM=4" "HIREST_M16_LM=5"; do
  echo "== $v"; env $v timeout 300 python tools/caption_profile.py 5 2>&1 | tail -1
done
