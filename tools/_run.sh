timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python tools/secondary_bench.py > gpurun_out/secondary_now.json 2> gpurun_out/secondary_now.err; tail -c 600 gpurun_out/secondary_now.err
