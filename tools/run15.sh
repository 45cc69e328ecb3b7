#!/bin/bash
mkdir -p gpurun_out/run15
timeout 600 python -m pytest tests/test_sentence_encoder.py -m gpu -q -x -s 2>&1 | tail -15
timeout 300 python tools/asr_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run15/asr_bench.txt
