#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "real_frames" -s 2>&1 | grep -E "real frames|passed|failed" | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "headline_config" -s 2>&1 | grep -E "headline config|passed|failed" | tail -20
