cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -k "handover_switches" 2>&1 | tail -8
