#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
B=streamflow_amd/csrc/build
for a in "gelu 2" "none 2" "relu 0" "none 0" "gelu 0"; do
  SF_SINGLE=1 SF_HIP_LIB=$B/variant_bst.so timeout 300 python tools/gemm_bs_timers.py 960 640 $a 2>&1 | grep -v amdgpu.ids | grep "single=\|epilogue\|whole\|mfma"
done > $O/s7_epi.log 2>&1; cat $O/s7_epi.log
