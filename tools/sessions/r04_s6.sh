#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
B=streamflow_amd/csrc/build
for v in bst bsts; do for single in 1 0; do
  echo "== $v"; SF_SINGLE=$single SF_HIP_LIB=$B/variant_$v.so timeout 300 python tools/gemm_bs_timers.py 960 640 2>&1 | grep -v amdgpu.ids
done; done > $O/s6_timers.log 2>&1; cat $O/s6_timers.log
export SF_SHAPES=960x640,640x640,384x256,256x384 SF_ALGO=2
for rep in 1 2; do for single in 1 0; do
  echo "== rep $rep single=$single packed"; SF_SINGLE=$single timeout 200 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M"
  echo "== rep $rep single=$single scalar"; SF_SINGLE=$single SF_HIP_LIB=$B/variant_bstn.so timeout 200 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M"
done; done > $O/s6_gelu.log 2>&1; cat $O/s6_gelu.log
