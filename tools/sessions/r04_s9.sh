#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
B=streamflow_amd/csrc/build
export SF_SHAPES=960x640,640x640,384x256,256x384 SF_ALGO=2
for rep in 1 2; do for single in 1 0; do
  echo "== rep $rep single=$single base"; SF_SINGLE=$single timeout 200 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M"
  echo "== rep $rep single=$single stagger 60"; SF_SINGLE=$single SF_HIP_LIB=$B/variant_stg1.so timeout 200 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M"
  echo "== rep $rep single=$single stagger 110"; SF_SINGLE=$single SF_HIP_LIB=$B/variant_stg2.so timeout 200 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M"
done; done > $O/s9_stagger.log 2>&1; cat $O/s9_stagger.log
