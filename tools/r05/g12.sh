#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ffn_pair.py -q -x 2>&1 | tail -4
timeout 300 python tools/ffn_pair_bench.py 1 1
timeout 300 python tools/ffn_pair_bench.py 2 2
