#!/bin/bash
# A/B of csrc/corr_blocked.hip variants on the fp16 corr-only steps: tools/corr16_ab.sh NAME ... ("base" = shipped library)
for v in "$@"; do
  if [ "$v" = base ]; then unset SF_HIP_LIB; else export SF_HIP_LIB=$PWD/streamflow_amd/csrc/build/variant_$v.so; fi
  for wl in sintel kitti; do
  timeout 100 python bench.py --corr-only --workload $wl --no-cpu-baseline 2>/dev/null </dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v $wl', 'ms/step %.3f' % d['ms_per_step'], 'frac %.3f' % (d['value']/8000), 'build %.1f us' % r['build']['avg_us'], 'lookup %.1f us' % r['lookup']['avg_us'])"
  done
done
