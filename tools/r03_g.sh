#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for cfg in "512 0" "4096 0" "4096 768" "512 768"; do set -- $cfg; echo "rep $rep WM2_MIN_M=$1 BD256=$2"; SF_SINGLE=1 SF_GEMM_BD_WM2_MIN_M=$1 SF_GEMM_BD256=$2 python tools/gemm_koct_bench.py koct 2>&1 | grep -E "^M (960|640) " | awk '{printf "%s %s %s us | ", $2, $4, $7} END {print ""}'; done; done
