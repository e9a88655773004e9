#!/bin/bash
mkdir -p gpurun_out/r05t
{
b() { python bench.py --steps 15 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print(round(d['value'],1), round(d['ms_per_step'],2), 'epe', d.get('epe_vs_oracle'), 'hard', (d.get('epe_hard_case') or {}).get('relative_to_flow'), 'clock', r.get('clock'), 'single', (d.get('single_clip') or {}).get('value'))
k=d.get('kernels',{})
for n in ('temporal_block','temporal_attn','layernorm','gma_flash'):
    if n in k: print('   ',n,k[n]['launches_per_step'],k[n]['ms_per_step'],k[n]['avg_us'])
"; }
echo "fused temporal block:"; b
echo "seven launches:"; SF_ENGINE_OPTS=temporal_block=0 b
echo "fused, again:"; b
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
} 2>&1 | tee gpurun_out/r05t/temporal_ab.txt
