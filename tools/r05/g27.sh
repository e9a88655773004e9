#!/bin/bash
mkdir -p gpurun_out/r05t
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "corr or lookup or blocked or clock or contract" 2>&1 | tail -3
for i in 1 2; do
for v in new old; do
  L=""; [ $v = old ] && L=streamflow_amd/csrc/build/variant_w2a.so
  SF_HIP_LIB=$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['roofline_corr']; k=d['kernels']
print('$v', round(d['value'],1), 'corr', c['frac'], 'build', c['build_gbps'], 'lookup', c['lookup_gbps'], k['corr_lookup']['avg_us'], 'us')"
done; done
} 2>&1 | tee gpurun_out/r05t/lookup_w2.txt
