#!/bin/bash
# round-4 final evidence: GPU suite, smoke, bench lines for every configuration quoted in DESIGN.md / README.md
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04f; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 600 python bench.py > $O/r04_bench_default.json 2> $O/bench_default.err; tail -c 400 $O/r04_bench_default.json; echo
N="--no-cpu-baseline"
timeout 300 python bench.py $N --workload kitti > $O/r04_bench_kitti.json 2>/dev/null
timeout 300 python bench.py $N --workload spring --clips 1 > $O/r04_bench_spring.json 2>/dev/null
timeout 300 python bench.py $N --clips 1 > $O/r04_bench_clip1.json 2>/dev/null
timeout 300 python bench.py $N --preset fp32_class > $O/r04_bench_fp32class.json 2>/dev/null
timeout 300 python bench.py $N --preset config2_fp16 > $O/r04_bench_config2_fp16.json 2>/dev/null
for f in kitti spring clip1 fp32class config2_fp16; do python -c "
import json,sys
d=json.loads(open('$O/r04_bench_$f.json').read().strip().splitlines()[-1])
print('$f', round(d['value'],1), 'ff/s', round(d['ms_per_step'],2), 'ms/step corr frac', d.get('roofline_corr',{}).get('frac'), 'enc', d.get('encoder_ms_per_clip'))"; done
