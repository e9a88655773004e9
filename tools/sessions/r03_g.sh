#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for cfg in 192 128; do for single in 1 0; do echo "rep $rep single=$single BD_MIN_M=$cfg"; SF_SINGLE=$single SF_GEMM_BD_MIN_M=$cfg python tools/gemm_koct_bench.py koct 2>&1 | grep -E "^M 128" | awk '{printf "%s/%s %s | ", $2, $4, $7} END {print ""}'; done; done; done
