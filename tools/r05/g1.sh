#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm_bstat.py -x -q 2>&1 | tail -4
timeout 1500 python tests/analysis/hard_case_ablation.py 11 12 13 21 31 32 > $O/hard_case_ablation.jsonl 2> $O/hard.err
tail -3 $O/hard.err
cat $O/hard_case_ablation.jsonl
