#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_joint.py -m gpu -q -x 2>&1 | tail -4
timeout 300 python tools/caption_profile.py 2>&1 | grep captioning
timeout 300 python tools/caption_profile.py 3 2>&1 | grep captioning
