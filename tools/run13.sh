#!/bin/bash
mkdir -p gpurun_out/run13
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/run13/pytest_gpu.log 2>&1; tail -6 gpurun_out/run13/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/run13/smoke.log 2>&1; tail -2 gpurun_out/run13/smoke.log
