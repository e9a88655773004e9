#!/bin/bash
# phase timers + round structure of gemm_bstat_gk at M960 K640 (one / two products) and M640 K640
mkdir -p gpurun_out/r05t
bash tools/build_variant.sh bst gemm_bstat.hip -DSF_BSTAT_TIMERS > /dev/null 2>&1
export SF_HIP_LIB=streamflow_amd/csrc/build/variant_bst.so
for s in 1 0; do
  SF_SINGLE=$s python tools/gemm_bs_timers.py 960 640
  SF_SINGLE=$s python tools/gemm_bs_timers.py 384 256
done 2>&1 | tee gpurun_out/r05t/timers.txt
