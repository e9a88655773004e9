cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04e
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "gma_flash" 2>&1 | tail -4
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04e/bench_v16.json 2>/dev/null; python - <<'P'
import json
d=json.loads(open('gpurun_out/r04e/bench_v16.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config2_fp16_mode']['value'], d['single_clip']['value'])
k=d['kernels']
for n in ('gemm','dwconv15','dwconv7','gma_flash','temporal_attn','layernorm'): print(n, round(k[n]['ms_per_step'],3), k[n]['launches_per_step'], round(k[n]['avg_us'] or 0,1))
P
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
