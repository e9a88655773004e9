#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/s17_tests.log; cat $O/s17_tests.log
