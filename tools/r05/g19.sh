#!/bin/bash
# A/B: two accumulation chains per tile in gemm_bstat_gk (SF_BSTAT_ALT_ACC)
mkdir -p gpurun_out/r05t
python -m streamflow_amd.build > /dev/null 2>&1
bash tools/build_variant.sh alt0 gemm_bstat.hip -DSF_BSTAT_ALT_ACC=0 > /dev/null 2>&1
{
for rep in 1 2; do
for s in 1 0; do
  echo "== single=$s  ALT_ACC=1"; SF_SINGLE=$s python tools/gemm_koct_bench.py koct 2>&1 | grep -v amdgpu.ids
  echo "== single=$s  ALT_ACC=0"; SF_HIP_LIB=streamflow_amd/csrc/build/variant_alt0.so SF_SINGLE=$s python tools/gemm_koct_bench.py koct 2>&1 | grep -v amdgpu.ids
done
done
timeout 900 python -m pytest tests/test_gpu_gemm_bstat.py tests/test_gpu_gemm.py -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ALT1', d['value'], d['ms_per_step'], d.get('epe_vs_oracle'))"
SF_HIP_LIB=streamflow_amd/csrc/build/variant_alt0.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ALT0', d['value'], d['ms_per_step'], d.get('epe_vs_oracle'))"
} 2>&1 | tee gpurun_out/r05t/altacc.txt
