#!/bin/bash
# round-4 session 5: epilogue rewrite of the activation-stationary GEMM: parity, timers, per-shape timing, bench, engine refactor tests
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm_bstat.py -m gpu -x -q 2>&1 | tail -5 > $O/s5_tests.log; cat $O/s5_tests.log
V=streamflow_amd/csrc/build/variant_bst.so
for single in 1 0; do for s in "960 640" "384 256"; do
  SF_SINGLE=$single SF_HIP_LIB=$V timeout 300 python tools/gemm_bs_timers.py $s 2>&1 | grep -v amdgpu.ids
done; done > $O/s5_timers.log 2>&1; cat $O/s5_timers.log
for single in 1 0; do echo "== koct bench single=$single algo=2"; SF_ALGO=2 SF_SINGLE=$single timeout 300 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M\|^sum"; done > $O/s5_koct.log 2>&1; cat $O/s5_koct.log
timeout 600 python bench.py --no-cpu-baseline --gemm-shapes > $O/s5_bench_shapes.json 2> $O/s5_bench_shapes.err; tail -3 $O/s5_bench_shapes.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/s5_bench_shapes.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('single_clip'), d.get('config2_fp16_mode',{}).get('value'))
k=d['kernels']
for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step'])[:28]: print('  ',n, v['launches_per_step'], round(v['ms_per_step'],3), v['avg_us'], v.get('tflops'))
PY
timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 > $O/s5_tests_all.log; cat $O/s5_tests_all.log
