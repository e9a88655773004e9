#!/bin/bash
# round-3 GPU session 1: blocked-volume tests, corr microbench (rows vs blocked) at the three shapes, short bench
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_corr_blocked.py -x -q > $O/s1_tests.log 2>&1; echo "blocked tests rc $?"; tail -15 $O/s1_tests.log
for w in sintel kitti spring; do timeout 300 python tools/corrb_bench.py $w > $O/s1_corrb_$w.log 2>&1; echo "corrb $w rc $?"; cat $O/s1_corrb_$w.log | tail -4; done
timeout 600 python bench.py --no-cpu-baseline > $O/s1_bench.json 2> $O/s1_bench.err; echo "bench rc $?"; tail -3 $O/s1_bench.err
python - <<PY
import json
d=json.load(open('$O/s1_bench.json'))
print('default', round(d['value'],1), round(d['ms_per_step'],2), d.get('roofline_corr'), d.get('epe_vs_oracle'))
print({k:(v['ms_per_step'], v['launches_per_step']) for k,v in d['kernels'].items()})
PY
