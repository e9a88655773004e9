cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04e
for v in base bd256; do
L=""; [ $v = bd256 ] && L="streamflow_amd/csrc/build/variant_bd256.so"
SF_HIP_LIB=$L timeout 900 python bench.py --no-cpu-baseline --gemm-shapes > gpurun_out/r04e/bench_$v.json 2>/dev/null; python - $v <<'P'
import json,sys
d=json.loads(open('gpurun_out/r04e/bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k=d['kernels']
print(sys.argv[1], round(d['value'],1), round(d['ms_per_step'],2), 'M256K384e5', round(k['gemm M256 K384 b24 e5']['avg_us'],1), 'gemm', round(k['gemm']['ms_per_step'],2))
P
done
