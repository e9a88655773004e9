#!/bin/bash
# single-clip configuration (bench.py --clips 1): engine option sweep
for o in "" "split_solo=0" "split_solo=4" "sk_tail_all=1" "gma_per_chain=0" "parallel_branches=0" "stored_auto_px=1" "head_pairs=1" "auto_split_k=0"; do
  SF_ENGINE_OPTS="$o" timeout 200 python bench.py --clips 1 --no-cpu-baseline --no-kernel-breakdown 2>/dev/null </dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$o]', round(d['value'],1), 'ff/s', round(d['ms_per_step'],2), 'ms')"
done
