#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05g; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; echo
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05g/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
print('epe', d['epe_vs_oracle']['value'], 'hard', d['epe_hard_case']['value'], d['epe_hard_case']['relative_to_flow'], d['epe_hard_case']['within_1e-3_of_max(1,flow)'])
print('c2fp16', d['config2_fp16_mode']['value'], 'fp32', d['fp32_class_mode']['value'], 'single', d['single_clip']['value'], 'f2f', d['frames_to_flows_per_sec']['value'])
r=d['roofline']; print({k:v for k,v in r.items() if k not in ('per_kernel','method','traffic_note')})
for k in r['per_kernel']: print(k['kernel'], k['avg_us'], k['frac_hbm'], k['frac_mfma'], k['bound'])
print(d['roofline_corr'])
k=d['kernels']
for n in ('gma_project_v','gma_flash','gemm','dwconv15','dwconv7','corr_build','corr_lookup'): print(n, k.get(n))
PY
