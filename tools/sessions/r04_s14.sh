#!/bin/bash
# round-4 session 14: full GPU suite + bench with the integrated kernels
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/s14_tests.log; cat $O/s14_tests.log
timeout 600 python bench.py --no-cpu-baseline --gemm-shapes > $O/s14_bench_shapes.json 2> $O/s14_bench_shapes.err; tail -2 $O/s14_bench_shapes.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/s14_bench_shapes.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('single_clip'), d.get('config2_fp16_mode',{}).get('value'), d.get('fp32_class_mode',{}).get('value'))
k=d['kernels']
for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step'])[:40]: print('  ',n, v['launches_per_step'], round(v['ms_per_step'],3), v['avg_us'], v.get('tflops'))
PY
