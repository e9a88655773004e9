#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm_bstat.py -m gpu -q 2>&1 | tail -8 > $O/s10_tests.log; cat $O/s10_tests.log
for single in 1 0; do echo "== koct bench single=$single algo=2"; SF_ALGO=2 SF_SINGLE=$single timeout 300 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M\|^sum"; done > $O/s10_koct.log 2>&1; cat $O/s10_koct.log
timeout 600 python bench.py --no-cpu-baseline --gemm-shapes > $O/s10_bench_shapes.json 2> $O/s10_bench_shapes.err; tail -3 $O/s10_bench_shapes.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/s10_bench_shapes.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('single_clip'), d.get('config2_fp16_mode',{}).get('value'), d.get('epe_vs_oracle'))
k=d['kernels']
for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step'])[:28]: print('  ',n, v['launches_per_step'], round(v['ms_per_step'],3), v['avg_us'], v.get('tflops'))
PY
