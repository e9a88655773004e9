#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_corr_blocked.py -x -q > $O/s5_tests.log 2>&1; echo "tests rc $?"; tail -4 $O/s5_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "headline or skblock or update or forward" > $O/s5_tests2.log 2>&1; echo "tests2 rc $?"; tail -4 $O/s5_tests2.log
for v in 0 1; do SF_GEMM_BDIRECT=$v timeout 600 python bench.py --no-cpu-baseline > $O/s5_bench_$v.json 2> $O/s5_bench_$v.err; python - <<PY
import json
d=json.load(open('$O/s5_bench_$v.json'))
print('bdirect=$v', round(d['value'],1), round(d['ms_per_step'],2), 'gemm ms', d['kernels']['gemm']['ms_per_step'], d.get('epe_vs_oracle',{}).get('value'))
PY
done
