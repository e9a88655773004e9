#!/bin/bash
# round-4 session 16: full GPU suite, smoke, default bench line (with the CPU baseline / multi-seed EPE)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/s16_tests.log; cat $O/s16_tests.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 900 python bench.py > $O/s16_bench_default.json 2> $O/s16_bench_default.err; tail -5 $O/s16_bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/s16_bench_default.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','epe_vs_oracle','epe_hard_case','cpu_baseline','single_clip','config2_fp16_mode','fp32_class_mode','roofline_corr'):
    print(k, json.dumps(d.get(k))[:700])
print('roofline', json.dumps({k:v for k,v in d['roofline'].items() if k not in ('method','traffic_note')})[:900])
PY
