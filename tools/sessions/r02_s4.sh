#!/bin/bash
# round-2 GPU session 4: bandwidth ceilings, corr-build store variants, flash EPE, Twins encoder tests
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
python tools/membw.py 2>/dev/null | tee $O/membw.log
( timeout 900 python -m pytest tests -m gpu -x -q -s -k "twins or real_frames or flash_mode or fp16_volume_mode or two_ranks" > $O/pytest_s4.log 2>&1; echo "pytest rc $?" >> $O/pytest_s4.log )
grep -E "EPE|passed|failed|rc |Error|error" $O/pytest_s4.log | tail -30
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/bench_s4_$name.json 2> $O/bench_s4_$name.err; python - <<PY
import json
try:
    d=json.load(open('$O/bench_s4_$name.json'))
    k=d['kernels']
    print('$name', round(d['value'],1), round(d['ms_per_step'],2), d.get('roofline_corr',{}).get('frac'), {x:(k[x]['ms_per_step']) for x in k if x in ('corr_build','corr_lookup','gma_flash')}, d.get('epe_vs_oracle',{}).get('value'))
except Exception as e: print('$name failed', e)
PY
}
run scalar_nt2 --no-cpu-baseline --steps 5
SF_HIP_LIB=streamflow_amd/csrc/build/variant_nt0.so run scalar_nt0 --no-cpu-baseline --steps 5
SF_CORR_VEC=1 run vec_nt2 --no-cpu-baseline --steps 5
SF_CORR_VEC=1 SF_HIP_LIB=streamflow_amd/csrc/build/variant_nt0.so run vec_nt0 --no-cpu-baseline --steps 5
run f32_scalar --no-cpu-baseline --steps 5 --corr-dtype f32
SF_GMA_MODE=flash SF_FLASH_QKP=1 run flash1_epe --cpu-runs 1
SF_GMA_MODE=flash SF_FLASH_QKP=2 run flash2_epe --cpu-runs 1
SF_GMA_MODE=flash SF_FLASH_QKP=1 run flash1_f16x2 --cpu-runs 1 --precision f16x2
