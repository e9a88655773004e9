cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04e
timeout 900 python bench.py --no-cpu-baseline --gemm-shapes > gpurun_out/r04e/bench_x2.json 2>/dev/null; python - <<'P'
import json
d=json.loads(open('gpurun_out/r04e/bench_x2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config2_fp16_mode']['value'], d['single_clip']['value'])
k=d['kernels']
f=lambda x: 0.0 if x is None else float(x)
rows=[(f(v.get('ms_per_step')),n,int(f(v.get('launches_per_step'))),f(v.get('avg_us')), f(v.get('gbps_algorithmic'))) for n,v in k.items()]
for r in sorted(rows,reverse=True)[:28]: print("  %7.3f ms  %-34s x%-4d %8.1f us  %7.0f GB/s"%r)
P
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
