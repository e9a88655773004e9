#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for cfg in "2 0" "0 0" "2 384" "2 512" "4 0"; do set -- $cfg
  echo -n "rep $rep SF_SPLIT_SOLO=$1 SF_GEMM_BD_MIN_WG=$2: "; SF_SPLIT_SOLO=$1 SF_GEMM_BD_MIN_WG=$2 python bench.py --clips 1 --no-cpu-baseline --no-kernel-breakdown --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"
done; done
