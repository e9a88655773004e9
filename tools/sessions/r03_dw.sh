#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for lib in "" dwtg2 dwtg8; do
  L=""; [ -n "$lib" ] && L=$PWD/streamflow_amd/csrc/build/variant_$lib.so
  echo "rep $rep variant [$lib]"
  for args in "640 24 7 f16x2 f16out" "324 24 15 f16x2 f16out single" "256 24 15 f16x2 f16out single" "128 24 15 f16x2 f16out single" "384 8 15 f16x2 f16out"; do
    SF_HIP_LIB=$L python tools/dwconv_one.py $args 2>&1 | tail -1 | awk '{printf "%s %s %s us | ", $1, $2, $5}'
  done; echo
done; done
