#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ffn_pair.py -x -q 2>&1 | tail -6
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_ffn_pair.py 2>&1 | tail -6
N="--no-cpu-baseline"
for o in 1 0; do SF_ENGINE_OPTS="ffn_pairs=$o" timeout 600 python bench.py $N --gemm-shapes > $O/bench_pairs$o.json 2>/dev/null; python - <<PY
import json
d=json.loads(open('$O/bench_pairs$o.json').read().strip().splitlines()[-1])
print('ffn_pairs=$o', round(d['value'],1), round(d['ms_per_step'],2), 'c2fp16', round(d.get('config2_fp16_mode',{}).get('value',0),1), 'single', round(d['single_clip']['value'],1))
k=d['kernels']
for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step'])[:14]: print('   ', n, v['launches_per_step'], v['ms_per_step'], v['avg_us'])
PY
done
