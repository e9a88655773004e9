#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05b; mkdir -p $O
SF_HIP_LIB=streamflow_amd/csrc/build/variant_r04eops.so SF_VARIANTS=fp32_class,config2_fp16,config2_mixed timeout 900 python tests/analysis/hard_case_ablation.py 11 12 13 21 > $O/hard_case_r04lib.jsonl 2> $O/hard.err
cat $O/hard_case_r04lib.jsonl
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json; echo
