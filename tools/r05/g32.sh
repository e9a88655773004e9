#!/bin/bash
{
for o in "" "head_pairs=0"; do
  echo "opts [$o]"; SF_ENGINE_OPTS=$o python bench.py --steps 5 --warmup 2 --no-kernel-breakdown --cpu-runs 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d.get('epe_vs_oracle')
print(round(d['value'],1), 'epe', [(s['seed'], s['clip'], round(s['epe_px'],6)) for s in e['samples']] if e else None)"
done
} 2>&1 | tee gpurun_out/r05t/head_pairs_epe3.txt
