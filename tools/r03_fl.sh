#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "flash" 2>&1 | tail -5
for st in 0 1; do
  SF_FLASH_STATS=$st python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stats $st', round(d['value'],1), d['kernels']['gma_flash']['ms_per_step'], d['kernels']['flash_pack_qk'], d['epe_vs_oracle'] if 'epe_vs_oracle' in d else '')"
done
