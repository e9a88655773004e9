cd "$GRAFT_REPO_ROOT"; timeout 600 python tools/encoder_bench.py 2>&1 | tail -40
