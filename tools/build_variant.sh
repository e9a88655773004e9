#!/bin/bash
# Build an alternative libstreamflow_hip with extra -D flags for one source file (A/B kernel experiments):
#   tools/build_variant.sh NAME corr.hip -DSF_CORR_NSTAGE=3     ->  streamflow_amd/csrc/build/variant_NAME.so
# and run with SF_HIP_LIB=streamflow_amd/csrc/build/variant_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
B=streamflow_amd/csrc/build
python -m streamflow_amd.build > /dev/null
timeout 600 hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c streamflow_amd/csrc/$src -o $B/variant_${name}_${src%.hip}.o
objs=""
for f in misc corr corr_blocked corr_blocked32 conv gemm gemm_split gemm_bstat ffn_pair sk_tail temporal mask_upsample attn encoder; do
  if [ "$f.hip" = "$src" ]; then objs="$objs $B/variant_${name}_${src%.hip}.o"; else objs="$objs $B/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $B/variant_$name.so
echo $B/variant_$name.so
