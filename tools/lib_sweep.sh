#!/bin/bash
# library variants on one box, inside the step: tools/lib_sweep.sh "<bench args>" base NAME ...   (variant_NAME.so from tools/build_variant.sh)
A="$1"; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset SF_HIP_LIB; else export SF_HIP_LIB=$PWD/streamflow_amd/csrc/build/variant_$v.so; fi
  timeout 300 python bench.py $A --no-cpu-baseline --no-kernel-breakdown 2>/dev/null </dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), 'ff/s', round(d['ms_per_step'],2), 'ms')"
done
