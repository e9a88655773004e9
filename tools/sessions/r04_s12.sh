#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
B=streamflow_amd/csrc/build
for single in 1 0; do for s in "960 640" "384 256"; do
  SF_SINGLE=$single SF_HIP_LIB=$B/variant_bst.so timeout 300 python tools/gemm_bs_timers.py $s gelu 2 2>&1 | grep -v amdgpu.ids
done; done > $O/s12_timers.log 2>&1; cat $O/s12_timers.log
