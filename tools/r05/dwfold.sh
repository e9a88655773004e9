#!/bin/bash
mkdir -p gpurun_out/r05t
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dwconv or skblock" 2>&1 | tail -3
for v in fold nofold foldall fold nofold; do
  L=""; [ $v = nofold ] && L=streamflow_amd/csrc/build/variant_nofold.so; [ $v = foldall ] && L=streamflow_amd/csrc/build/variant_foldall.so
  SF_HIP_LIB=$L python bench.py --steps 15 --warmup 3 --cpu-runs 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; e=d.get('epe_vs_oracle'); hc=d.get('epe_hard_case')
print('$v', round(d['value'],1), round(d['ms_per_step'],2), 'dw15', k['dwconv15']['ms_per_step'], 'dw7', k['dwconv7']['ms_per_step'], 'epe', round(e['value'],6) if e else None, 'hard', round(hc['relative_to_flow'],6) if hc else None)"
done
} 2>&1 | tee gpurun_out/r05t/dw_fold.txt
