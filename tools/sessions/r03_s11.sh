#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -k "window_attn or subsample or koct_handover" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corr_blocked.py -x -q -k "twins or real_frames or integration_md" 2>&1 | tail -8
timeout 600 python tools/encoder_bench.py 1 shapes 2>&1 | tail -1 | cut -c1-1500
timeout 600 python tools/encoder_bench.py 4 2>&1 | tail -1 | cut -c1-600
