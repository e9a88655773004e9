#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c; mkdir -p $O
(cd _wt_r04pre && timeout 600 python hc.py > ../$O/hard_case_88f9fb4.jsonl 2> ../$O/hc.err); cat $O/hard_case_88f9fb4.jsonl; tail -2 $O/hc.err
N="--no-cpu-baseline"
timeout 600 python bench.py $N --gemm-shapes > $O/bench_shapes.json 2> $O/bench_shapes.err; tail -c 300 $O/bench_shapes.json; echo
timeout 600 python bench.py $N --gemm-shapes --preset fp32_class > $O/bench_fp32_shapes.json 2> $O/bench_fp32_shapes.err; tail -c 300 $O/bench_fp32_shapes.json; echo
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "hard_case_sweep" 2>&1 | tail -5
