#!/bin/bash
# where do the half-batch chains (EngineOptions.split_solo) start to pay?  clips per step x split_solo
for c in 1 2 3 4 6; do for o in "split_solo=0" "split_solo=2"; do
  SF_ENGINE_OPTS="$o" timeout 200 python bench.py --clips $c --no-cpu-baseline --no-kernel-breakdown 2>/dev/null </dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('clips $c [$o]', round(d['value'],1), 'ff/s', round(d['ms_per_step'],2), 'ms')"
done; done
for o in "split_solo=0" "split_solo=2"; do
  SF_ENGINE_OPTS="$o" timeout 200 python bench.py --workload kitti --no-cpu-baseline --no-kernel-breakdown 2>/dev/null </dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('kitti 8 clips [$o]', round(d['value'],1), 'ff/s', round(d['ms_per_step'],2), 'ms')"
  SF_ENGINE_OPTS="$o" timeout 200 python bench.py --workload spring --clips 1 --no-cpu-baseline --no-kernel-breakdown 2>/dev/null </dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('spring 1 clip [$o]', round(d['value'],1), 'ff/s', round(d['ms_per_step'],2), 'ms')"
done
