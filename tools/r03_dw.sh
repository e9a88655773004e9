#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for args in "640 24 7 f16x2 f16out" "324 24 15 f16x2 f16out single" "256 24 15 f16x2 f16out single" "384 8 15 f16x2 f16out"; do
  SF_DW_TS=1 SF_HIP_LIB=$PWD/streamflow_amd/csrc/build/variant_dwt.so python tools/dwconv_one.py $args 2>&1 | tail -3
done
