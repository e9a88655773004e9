#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/s4_gputests.log 2>&1; echo "gpu tests rc $?"; tail -5 $O/s4_gputests.log
timeout 900 python bench.py > $O/s4_bench.json 2> $O/s4_bench.err; echo "bench rc $?"; tail -2 $O/s4_bench.err
python - <<PY
import json
d=json.load(open('$O/s4_bench.json'))
print('default', round(d['value'],1), round(d['ms_per_step'],2), d.get('roofline_corr'), d.get('epe_vs_oracle'))
print({k:(v['ms_per_step'], v['launches_per_step']) for k,v in d['kernels'].items()})
PY
