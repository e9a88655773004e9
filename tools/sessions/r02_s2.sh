#!/bin/bash
# round-2 GPU session 2: fp16 correlation volumes (tests + bench), GEMM fragment-prefetch variant A/B
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q -k "fp16 or corr or two_ranks or bilinear" > $O/pytest_s2.log 2>&1; echo "pytest rc $?" >> $O/pytest_s2.log )
tail -15 $O/pytest_s2.log
for cd in f16 f32; do
  timeout 600 python bench.py --corr-dtype $cd --cpu-runs 1 > $O/bench_s2_$cd.json 2> $O/bench_s2_$cd.err; echo "bench $cd rc $?"
  python - <<PY
import json
d=json.load(open('$O/bench_s2_$cd.json'))
print('$cd', d['value'], d['ms_per_step'], d.get('roofline_corr'), d.get('epe_vs_oracle',{}).get('value'), {k:(v['ms_per_step'],v['gbps_algorithmic']) for k,v in d['kernels'].items() if 'corr' in k})
PY
done
timeout 300 python bench.py --workload kitti --no-cpu-baseline > $O/bench_s2_kitti.json 2>/dev/null
python -c "
import json
d=json.load(open('$O/bench_s2_kitti.json'));print('kitti', d['value'], d.get('roofline_corr'))"
V=streamflow_amd/csrc/build/variant_pf.so
for shape in "960 640 gelu" "640 960 none" "384 256 gelu" "256 384 none" "128 960 none"; do
  for r in 1 2; do
    SF_N_IMG=24 python tools/gemm_one.py $shape 2>/dev/null | tail -1 | sed 's/^/base /'
    SF_N_IMG=24 SF_HIP_LIB=$V python tools/gemm_one.py $shape 2>/dev/null | tail -1 | sed 's/^/pf   /'
  done
done
SF_HIP_LIB=$V timeout 300 python bench.py --no-cpu-baseline > $O/bench_s2_pf.json 2>/dev/null
python -c "
import json
d=json.load(open('$O/bench_s2_pf.json'));print('pf bench', d['value'], d['ms_per_step'], d['kernels']['gemm'])"
