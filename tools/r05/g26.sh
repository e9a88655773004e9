#!/bin/bash
b() { python bench.py --steps 30 --warmup 5 --clips 1 --no-cpu-baseline --no-kernel-breakdown 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in 1 2; do
echo "single clip, default:"; b
echo "single clip, ffn2 pairs always:"; SF_EXP_PAIR2_SMALL=1 b
done
