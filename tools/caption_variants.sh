#!/bin/bash
# captions/s for tuning variants of the decode GEMMs (environment switches read once per process)
for v in "" "HIREST_M16_LN=1" "HIREST_M16_LN=2" "HIREST_M16_LN=3" "HIREST_M16_MID=1" "HIREST_M16_MID=2" "HIREST_M16_LM=1" "HIREST_M16_LM=2" ""; do
  echo "== $v"; env $v timeout 300 python tools/caption_profile.py 5 2>&1 | tail -1
done
