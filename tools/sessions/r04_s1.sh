#!/bin/bash
# round-4 session 1: where does a stage of the B-direct GEMM go? (phase timers) + per-shape baseline on this box
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
V=streamflow_amd/csrc/build/variant_bdt.so
for single in 1 0; do for s in "960 640" "640 960" "384 256"; do
  echo "== timers single=$single M K = $s"; SF_SINGLE=$single SF_HIP_LIB=$V timeout 300 python tools/gemm_bd_timers.py $s 2>&1 | tail -9
done; done > $O/s1_timers.log 2>&1
for single in 1 0; do echo "== koct bench single=$single"; SF_SINGLE=$single timeout 300 python tools/gemm_koct_bench.py koct 2>&1 | tail -14; done > $O/s1_koct.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --gemm-shapes > $O/s1_bench_shapes.json 2> $O/s1_bench_shapes.err
timeout 600 python bench.py --no-cpu-baseline --gemm-shapes --preset config2_fp16 > $O/s1_bench_shapes_fp16.json 2> $O/s1_bench_shapes_fp16.err
tail -c 300 $O/s1_bench_shapes.json
