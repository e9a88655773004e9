#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "stress_shape or debug_range or spring_shape_smoke" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -k "gelu or koct_rows" 2>&1 | tail -3
