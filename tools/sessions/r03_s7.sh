#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "preset_selected or single_product or headline" 2>&1 | tail -4
timeout 900 python bench.py --no-cpu-baseline > $O/s7_bench.json 2> $O/s7_bench.err; echo "bench rc $?"; tail -2 $O/s7_bench.err
python - <<PY
import json
d=json.load(open('$O/s7_bench.json'))
print('default', d['config']['preset'], round(d['value'],1), round(d['ms_per_step'],2), d['roofline'].get('frac'), d['roofline'].get('frac_mfma'), d.get('roofline_corr',{}).get('frac'))
print({k:(v['ms_per_step'], v['launches_per_step']) for k,v in d['kernels'].items()})
for k in ('config2_fp16_mode','fp32_class_mode','single_clip'): print(k, d.get(k))
PY
