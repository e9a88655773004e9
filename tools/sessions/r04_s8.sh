#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
B=streamflow_amd/csrc/build
timeout 900 python -m pytest tests/test_gpu_gemm_bstat.py -m gpu -x -q 2>&1 | tail -5 > $O/s8_tests.log; cat $O/s8_tests.log
for a in "gelu 2" "gelu 0"; do for single in 1 0; do
  SF_SINGLE=$single SF_HIP_LIB=$B/variant_bst.so timeout 300 python tools/gemm_bs_timers.py 960 640 $a 2>&1 | grep -v amdgpu.ids | grep "single=\|epilogue\|whole\|mfma"
done; done > $O/s8_epi.log 2>&1; cat $O/s8_epi.log
for single in 1 0; do echo "== koct bench single=$single algo=2"; SF_ALGO=2 SF_SINGLE=$single timeout 300 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M\|^sum"; done > $O/s8_koct.log 2>&1; cat $O/s8_koct.log
timeout 600 python bench.py --no-cpu-baseline --gemm-shapes > $O/s8_bench_shapes.json 2> $O/s8_bench_shapes.err; tail -3 $O/s8_bench_shapes.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/s8_bench_shapes.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('single_clip'), d.get('config2_fp16_mode',{}).get('value'), d.get('epe_vs_oracle'))
k=d['kernels']
for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step'])[:28]: print('  ',n, v['launches_per_step'], round(v['ms_per_step'],3), v['avg_us'], v.get('tflops'))
PY
