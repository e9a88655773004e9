cd "$GRAFT_REPO_ROOT"
for n in 6 12 24 48 96; do echo "== n_img $n"; SF_N_IMG=$n SF_SINGLE=1 SF_SHAPES=384x256,960x640 timeout 300 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M"; done
