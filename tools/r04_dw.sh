cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "bench_strong or bench_two" 2>&1 | tail -8
