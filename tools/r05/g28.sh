#!/bin/bash
mkdir -p gpurun_out/r05t
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ffn_pair.py -x -q -m gpu -k "flash or gma" 2>&1 | tail -3
for i in 1 2; do
for v in new old; do
  L=""; [ $v = old ] && L=streamflow_amd/csrc/build/variant_noxcd.so
  SF_HIP_LIB=$L python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$v', round(d['value'],1), round(d['ms_per_step'],2), 'gma', k['gma_flash']['avg_us'], 'us', 'clock', d['roofline']['clock']['sustained_mhz'])"
done; done
} 2>&1 | tee gpurun_out/r05t/flash_xcd.txt
