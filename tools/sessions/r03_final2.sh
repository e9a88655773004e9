#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03f; mkdir -p $O
timeout 600 python bench.py > $O/r03_bench_default.json 2> $O/bench_default.err
N="--no-cpu-baseline"
timeout 300 python bench.py $N --workload kitti > $O/r03_bench_kitti.json 2>/dev/null
timeout 300 python bench.py $N --workload spring --clips 1 > $O/r03_bench_spring.json 2>/dev/null
timeout 300 python bench.py $N --clips 1 > $O/r03_bench_clip1.json 2>/dev/null
for f in default kitti spring clip1; do python -c "
import json,sys
d=json.loads(open('$O/r03_bench_$f.json').read().strip().splitlines()[-1])
print('$f', round(d['value'],1), 'ff/s', round(d['ms_per_step'],2), 'ms/step corr frac', d.get('roofline_corr',{}).get('frac'), 'enc', d.get('encoder_ms_per_clip'), 'single', d.get('single_clip',{}).get('value'))"; done
