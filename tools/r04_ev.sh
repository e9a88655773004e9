cd "$GRAFT_REPO_ROOT"; timeout 1200 python -m pytest tests/test_gpu_evaluate.py -x -q -s 2>&1 | tail -25
