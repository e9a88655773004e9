cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_gemm_bstat.py -q -x 2>&1 | tail -2
SF_SINGLE=1 SF_SHAPES=256x486,486x324,384x256 timeout 300 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M"
SF_SHAPES=256x486 timeout 300 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M"
