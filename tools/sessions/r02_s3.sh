#!/bin/bash
# round-2 GPU session 3: vector epilogue of the build, new fp16 lookup, flash GMA kernel
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -x -q -k "fp16 or corr or flash or sintel or kitti or spring or golden" > $O/pytest_s3.log 2>&1; echo "pytest rc $?" >> $O/pytest_s3.log )
grep -E "flash|fp16 vol|passed|failed|rc" $O/pytest_s3.log | tail -30
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > $O/bench_s3_$name.json 2> $O/bench_s3_$name.err; python - <<PY
import json
try:
    d=json.load(open('$O/bench_s3_$name.json'))
    k=d['kernels']
    print('$name', round(d['value'],1), round(d['ms_per_step'],2), d.get('roofline_corr',{}).get('frac'), {x:(k[x]['ms_per_step'],k[x]['avg_us']) for x in k if x in ('corr_build','corr_lookup','gma_flash','gemm_attn','softmax_rows','flash_pack_qk')}, d.get('single_clip',{}).get('value'))
except Exception as e: print('$name failed', e)
PY
}
run f16 --corr-dtype f16
run f32 --corr-dtype f32
SF_GMA_MODE=flash SF_FLASH_QKP=1 run flash1
SF_GMA_MODE=flash SF_FLASH_QKP=2 run flash2
SF_GMA_MODE=flash SF_FLASH_QKP=3 run flash3
run kitti --workload kitti
run spring --workload spring --clips 1 --steps 5
SF_FLASH_QKP=1 run spring1 --workload spring --clips 1 --steps 5
SF_GMA_MODE=matrix run springm --workload spring --clips 1 --steps 5
