cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04e
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
timeout 900 python bench.py --gemm-shapes > gpurun_out/r04e/bench_kio.json 2>/dev/null; python - <<'P'
import json
d=json.loads(open('gpurun_out/r04e/bench_kio.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config2_fp16_mode']['value'], d['single_clip']['value'])
print([round(s['epe_px']*1e4,2) for s in d['epe_vs_oracle']['samples']], 'hard', round(d['epe_hard_case']['value'],4))
k=d['kernels']
f=lambda x: 0.0 if x is None else float(x)
rows=[(f(v.get('ms_per_step')),n,int(f(v.get('launches_per_step'))),f(v.get('avg_us')), f(v.get('gbps_algorithmic'))) for n,v in k.items()]
for r in sorted(rows,reverse=True)[:40]: print("  %7.3f ms  %-34s x%-4d %8.1f us  %7.0f GB/s"%r)
P
