#!/bin/bash
mkdir -p gpurun_out/r05t
{
python tools/clock_probe_check2.py 2>&1 | grep -v amdgpu.ids | tail -8
timeout 1500 python -m pytest tests/test_gpu_ffn_pair.py -x -q -m gpu 2>&1 | tail -5
b() { python bench.py --steps 15 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print(round(d['value'],1), round(d['ms_per_step'],2), 'epe', d.get('epe_vs_oracle'), 'hard', (d.get('epe_hard_case') or {}).get('relative_to_flow'), 'single', (d.get('single_clip') or {}).get('value'))
k=d.get('kernels',{})
for n in sorted(k):
    if 'b8' in n or 'M384' in n or 'M6 ' in n: print('   ',n,k[n]['launches_per_step'],k[n]['ms_per_step'],k[n]['avg_us'])
"; }
echo "flow-head pairs:"; b --gemm-shapes
echo "again:"; b
} 2>&1 | tee gpurun_out/r05t/flowhead_pairs.txt
