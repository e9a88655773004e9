#!/bin/bash
# round-4 session 2: the activation-stationary GEMM: parity tests, per-shape timing against the tiled kernels, bench
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm_bstat.py -m gpu -x -q 2>&1 | tail -15 > $O/s2_tests.log
cat $O/s2_tests.log
for single in 1 0; do for algo in 1 2; do echo "== koct bench single=$single algo=$algo"; SF_ALGO=$algo SF_SINGLE=$single timeout 300 python tools/gemm_koct_bench.py koct 2>&1 | tail -13; done; done > $O/s2_koct.log 2>&1
cat $O/s2_koct.log
timeout 600 python bench.py --no-cpu-baseline --gemm-shapes > $O/s2_bench_shapes.json 2> $O/s2_bench_shapes.err
tail -c 600 $O/s2_bench_shapes.json; tail -3 $O/s2_bench_shapes.err
