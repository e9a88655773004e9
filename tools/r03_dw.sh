#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for args in "640 24 7 f16x2 f16out" "324 24 15 f16x2 f16out single"; do
  python tools/dwconv_one.py $args 2>&1 | tail -1
  for a in 1 2 3; do echo "ablate $a (1 no stores, 2 no loads, 3 one kernel row)"; SF_HIP_LIB=$PWD/streamflow_amd/csrc/build/variant_dwab$a.so python tools/dwconv_one.py $args 2>&1 | tail -1; done
done
