#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
for rep in 1 2; do SF_SHAPES=384x256,256x256,384x384 SF_ALGO=2 SF_SINGLE=0 timeout 200 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M"; done > $O/s11_anom.log 2>&1; cat $O/s11_anom.log
export SF_SHAPES=256x384,128x192,384x256,192x128,640x960
for single in 1 0; do for algo in 1 2; do echo "== dw1 single=$single algo=$algo"; SF_ALGO=$algo SF_SINGLE=$single timeout 200 python tools/gemm_koct_bench.py dw1 2>&1 | grep "^M"; done; done > $O/s11_dw1.log 2>&1; cat $O/s11_dw1.log
for single in 1 0; do for algo in 1 2; do echo "== none(fp32 out) single=$single algo=$algo"; SF_ALGO=$algo SF_SINGLE=$single timeout 200 python tools/gemm_koct_bench.py none 2>&1 | grep "^M"; done; done > $O/s11_none.log 2>&1; cat $O/s11_none.log
