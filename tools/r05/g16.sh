#!/bin/bash
cd "$GRAFT_REPO_ROOT"
SF_HIP_LIB=streamflow_amd/csrc/build/variant_pair_one.so timeout 900 python -m pytest tests/test_gpu_ffn_pair.py -q -x 2>&1 | tail -3
for i in 1 2; do
echo "-- default"; timeout 300 python tools/ffn_pair_bench.py 1 1 | head -3
echo "-- one loader"; SF_HIP_LIB=streamflow_amd/csrc/build/variant_pair_one.so timeout 300 python tools/ffn_pair_bench.py 1 1 | head -3
done
SF_HIP_LIB=streamflow_amd/csrc/build/variant_pair_timers.so timeout 120 python tools/ffn_pair_timers.py 1 256 384 256 1 1
SF_HIP_LIB=streamflow_amd/csrc/build/variant_pair_one_t.so timeout 120 python tools/ffn_pair_timers.py 1 256 384 256 1 1
