#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_corr_blocked.py -q -k "gemm or koct" 2>&1 | tail -2
for i in 1 2; do
for wm in 512 -1; do
  SF_GEMM_BD_WM2_MIN_M=$wm python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wm2_min $wm', round(d['value'],1), d['kernels']['gemm']['ms_per_step'])"
done; done
