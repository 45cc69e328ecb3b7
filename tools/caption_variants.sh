#!/bin/bash
# captions/s for tuning variants of the decode GEMMs (environment switches read once per process)
timeout 600 python -m pytest tests/test_gpu_joint.py -m gpu -q -x -k "caption or gemm_f32" 2>&1 | tail -2
for v in "" "HIREST_M16_LN=1" "HIREST_M16_LN=2" ""; do
  echo "== $v"; env $v timeout 300 python tools/caption_profile.py 5 2>&1 | tail -1
done
