#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05d; mkdir -p $O
timeout 2400 python tests/analysis/preset_select_v2.py > $O/r05_preset_select.jsonl 2> $O/sel.err; tail -3 $O/sel.err; tail -c 2500 $O/r05_preset_select.jsonl
N="--no-cpu-baseline --no-kernel-breakdown"
for q in 1 3; do timeout 600 python bench.py $N --preset fp32_class --gma flash --flash-qkp $q > $O/bench_fp32_flash$q.json 2>/dev/null; python -c "
import json;d=json.loads(open('$O/bench_fp32_flash$q.json').read().strip().splitlines()[-1]);print('fp32_class flash qkp $q', d['value'], d['ms_per_step'])"; done
