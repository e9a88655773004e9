#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -k "window_attn" -s 2>&1 | grep -E "window_attn|passed|failed|Error|assert" | cut -c1-200 | tail -24
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "twins or real_frames" 2>&1 | tail -4
timeout 600 python tools/encoder_bench.py 1 shapes 2>&1 | tail -2 | cut -c1-1500
