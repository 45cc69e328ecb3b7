#!/bin/bash
# does the clock policy explain the gap between profiled and unprofiled step-captioning times?
rocm-smi --showperflevel --showclocks 2>&1 | grep -i "perf\|sclk\|mclk" | head -6
timeout 300 python tools/caption_profile.py 5 2>&1 | tail -1
( timeout 60 python tools/caption_profile.py 5 > /dev/null 2>&1 & sleep 25; rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | head -4; wait )
rocm-smi --setperflevel high 2>&1 | tail -2
rocm-smi --showperflevel --showclocks 2>&1 | grep -i "perf\|sclk" | head -4
timeout 300 python tools/caption_profile.py 5 2>&1 | tail -1
rocm-smi --setperflevel auto 2>&1 | tail -1
