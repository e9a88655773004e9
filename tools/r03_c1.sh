#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for t in 0 384 512 768 1024; do
  echo "SF_GEMM_BD_MIN_WG=$t"; SF_GEMM_BD_MIN_WG=$t python bench.py --clips 1 --no-cpu-baseline --no-kernel-breakdown --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
