#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json; tail -3 $O/bench_default.err
