#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export SF_HIP_LIB=streamflow_amd/csrc/build/variant_pair_timers.so
timeout 120 python tools/ffn_pair_timers.py 1 256 384 256 1 1
timeout 120 python tools/ffn_pair_timers.py 1 256 384 256 2 2
timeout 120 python tools/ffn_pair_timers.py 0 256 384 192 1 1
timeout 120 python tools/ffn_pair_timers.py 0 324 486 256 2 1
timeout 120 python tools/ffn_pair_timers.py 1 324 486 324 1 1
