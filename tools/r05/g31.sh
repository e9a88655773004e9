#!/bin/bash
mkdir -p gpurun_out/r05t
{
timeout 900 python -m pytest tests/test_gpu_ffn_pair.py -x -q -m gpu 2>&1 | tail -4
for o in "" "head_pairs=0"; do
  echo "opts [$o]"; SF_ENGINE_OPTS=$o python bench.py --steps 10 --warmup 3 --cpu-runs 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d.get('epe_vs_oracle')
print(round(d['value'],1), round(d['ms_per_step'],2), 'epe', [round(s['epe_px'],6) for s in e['samples']] if e else None)"
done
} 2>&1 | tee gpurun_out/r05t/head_pairs_r32.txt
