#!/bin/bash
# engine option sweep on one box: tools/opt_sweep.sh "<bench args>" opt1 opt2 ...   ("" = defaults)
A="$1"; shift
for o in "$@"; do
  SF_ENGINE_OPTS="$o" timeout 300 python bench.py $A --no-cpu-baseline --no-kernel-breakdown 2>/dev/null </dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$o]', round(d['value'],1), 'ff/s', round(d['ms_per_step'],2), 'ms')"
done
