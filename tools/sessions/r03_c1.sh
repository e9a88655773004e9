#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for so in 2 0; do
  echo -n "rep $rep SF_SPLIT_SOLO=$so: "; SF_SPLIT_SOLO=$so python bench.py --clips 1 --no-cpu-baseline --no-kernel-breakdown --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"
done; done
