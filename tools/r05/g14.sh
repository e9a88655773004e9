#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ffn_pair.py -q -x 2>&1 | tail -3
SF_HIP_LIB=streamflow_amd/csrc/build/variant_pair_sp2.so timeout 900 python -m pytest tests/test_gpu_ffn_pair.py -q -x 2>&1 | tail -3
for i in 1 2; do
echo "-- spread 0"; timeout 300 python tools/ffn_pair_bench.py 1 1
echo "-- spread 2"; SF_HIP_LIB=streamflow_amd/csrc/build/variant_pair_sp2.so timeout 300 python tools/ffn_pair_bench.py 1 1
done
echo "-- spread 2, (2,2)"; SF_HIP_LIB=streamflow_amd/csrc/build/variant_pair_sp2.so timeout 300 python tools/ffn_pair_bench.py 2 2
export SF_HIP_LIB=streamflow_amd/csrc/build/variant_pair_timers.so
timeout 120 python tools/ffn_pair_timers.py 1 256 384 256 1 1
timeout 120 python tools/ffn_pair_timers.py 0 256 384 192 1 1
timeout 120 python tools/ffn_pair_timers.py 1 324 486 324 1 1
