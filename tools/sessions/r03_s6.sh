#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
for v in 0 1; do SF_GEMM_BDIRECT=$v timeout 600 python bench.py --no-cpu-baseline --gemm-shapes > $O/s6_bench_$v.json 2> $O/s6_bench_$v.err; python - <<PY
import json
d=json.load(open('$O/s6_bench_$v.json'))
print('bdirect=$v', round(d['value'],1), round(d['ms_per_step'],2), d.get('epe_vs_oracle',{}).get('value'))
rows=[(k,v['ms_per_step'],v['launches_per_step'],v.get('tflops')) for k,v in d['kernels'].items()]
rows.sort(key=lambda r:-r[1])
tot=sum(r[1] for r in rows if r[0].startswith('gemm'))
print('gemm total', round(tot,2))
for r in rows[:26]: print('   ', r)
PY
done
