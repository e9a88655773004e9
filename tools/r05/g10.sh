#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for v in "" streamflow_amd/csrc/build/variant_pair_safe.so streamflow_amd/csrc/build/variant_pair_r2.so; do
  echo "== lib: ${v:-default}"
  SF_HIP_LIB=$v timeout 600 python -m pytest tests/test_gpu_ffn_pair.py -q -k "deterministic" 2>&1 | tail -7
done
timeout 300 python tools/ffn_pair_bench.py 1 1
SF_HIP_LIB=streamflow_amd/csrc/build/variant_pair_r2.so timeout 300 python tools/ffn_pair_bench.py 1 1
