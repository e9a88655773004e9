#!/bin/bash
mkdir -p gpurun_out/r05t
{
timeout 900 python -m pytest tests/test_gpu_mask_upsample.py -x -q -m gpu 2>&1 | tail -4
for o in "" "mask_upsample=0" ""; do echo "opts [$o]"; SF_ENGINE_OPTS=$o python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print(round(d['value'],1), round(d['ms_per_step'],2), {n:(k[n]['launches_per_step'],k[n]['avg_us']) for n in k if 'mask' in n or 'upsample' in n})"; done
python bench.py --steps 10 --warmup 3 --clips 1 --no-cpu-baseline --no-kernel-breakdown 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('clip1', round(d['value'],1))"
} 2>&1 | tee gpurun_out/r05t/mask_upsample.txt
