#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "flash" 2>&1 | tail -3
for rep in 1 2; do
for sp in 1 -1; do
  echo -n "rep $rep SF_FLASH_SPLIT=$sp: "; SF_FLASH_SPLIT=$sp python bench.py --clips 1 --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), d['kernels']['gma_flash'])"
done; done
