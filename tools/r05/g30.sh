#!/bin/bash
mkdir -p gpurun_out/r05t
{
for o in "" "temporal_block=0" "head_pairs=0" "temporal_block=0,head_pairs=0,mask_upsample=0"; do
  echo "opts [$o]"; SF_ENGINE_OPTS=$o python bench.py --steps 5 --warmup 2 --cpu-runs 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d.get('epe_vs_oracle'); hc=d.get('epe_hard_case')
print(round(d['value'],1), 'epe', [round(s['epe_px'],6) for s in e['samples']] if e else None, 'hard', [(s['seed'], round(s['relative_to_flow'],6)) for s in hc['seeds']] if hc else None)"
done
} 2>&1 | tee gpurun_out/r05t/epe_budget.txt
