#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for bd in 0 1; do echo "SF_GEMM_BD256=$bd single koct"; SF_SINGLE=1 SF_GEMM_BD256=$bd python tools/gemm_koct_bench.py koct 2>&1 | grep -E "^M|sum"; done
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_corr_blocked.py -q -k "gemm or koct" 2>&1 | tail -3
