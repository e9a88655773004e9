#!/bin/bash
# round-6 final evidence: GPU suite, smoke, bench lines for every configuration quoted in DESIGN.md / README.md
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06f; mkdir -p $O
# (when tools/r06/profiles.sh ran earlier in the same call: bench.py quotes PMC traffic only from files taken on these sources)
cp gpurun_out/r06p/r06_traffic_*.json gpurun_out/r06p/r06_mfma_busy_*.json profiles/ 2>/dev/null
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 900 python bench.py > $O/r06_bench_default.json 2> $O/bench_default.err; tail -c 400 $O/r06_bench_default.json; echo
N="--no-cpu-baseline"
timeout 300 python bench.py $N --workload kitti > $O/r06_bench_kitti.json 2>/dev/null
timeout 300 python bench.py $N --workload spring --clips 1 > $O/r06_bench_spring.json 2>/dev/null
timeout 300 python bench.py $N --clips 1 > $O/r06_bench_clip1.json 2>/dev/null
timeout 300 python bench.py $N --preset fp32_class > $O/r06_bench_fp32class.json 2>/dev/null
timeout 300 python bench.py $N --preset config2_fp16 > $O/r06_bench_config2_fp16.json 2>/dev/null
timeout 300 python bench.py --corr-only --workload kitti --preset fp32_class > $O/r06_bench_kitti_fp32_corr.json 2>/dev/null
timeout 300 python bench.py --corr-only --workload kitti --preset fp32_class --corr-layout rows $N > $O/r06_bench_kitti_fp32_corr_rows.json 2>/dev/null
timeout 300 python bench.py --corr-only --workload sintel --preset fp32_class $N > $O/r06_bench_sintel_fp32_corr.json 2>/dev/null
timeout 300 python bench.py --corr-only --workload sintel --preset fp32_class --corr-layout rows $N > $O/r06_bench_sintel_fp32_corr_rows.json 2>/dev/null
timeout 300 python bench.py --corr-only --workload sintel $N > $O/r06_bench_sintel_fp16_corr.json 2>/dev/null
timeout 300 python bench.py $N --preset fp32_class --workload spring --clips 1 > $O/r06_bench_spring_fp32class.json 2>/dev/null
timeout 300 python bench.py --corr-only --workload kitti > $O/r06_bench_kitti_fp16_corr.json 2>/dev/null
timeout 300 python bench.py $N --gpus 2 --share-device --dist-backend gloo --steps 5 > $O/r06_bench_2ranks_shared.json 2>/dev/null
timeout 600 python bench.py $N --no-kernel-breakdown --gpus 8 --share-device --dist-backend gloo --steps 3 --warmup 1 > $O/r06_bench_8ranks_shared.json 2>/dev/null
timeout 300 python bench.py $N --gma stored > $O/r06_bench_gma_stored.json 2>/dev/null
for f in kitti spring clip1 fp32class spring_fp32class config2_fp16 gma_stored; do python -c "
import json,sys
d=json.loads(open('$O/r06_bench_$f.json').read().strip().splitlines()[-1])
print('$f', round(d['value'],1), 'ff/s', round(d['ms_per_step'],2), 'ms/step corr frac', d.get('roofline_corr',{}).get('frac'), 'enc', d.get('encoder_ms_per_clip'))"; done
for f in kitti_fp32_corr kitti_fp32_corr_rows sintel_fp32_corr sintel_fp32_corr_rows kitti_fp16_corr sintel_fp16_corr; do python -c "
import json
d=json.loads(open('$O/r06_bench_$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', round(r['frac'],4), 'build', round(r['build']['avg_us'],1), round(r['build']['gbps'],0), 'lookup', round(r['lookup']['avg_us'],1), round(r['lookup']['gbps'],0))"; done
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06f/r06_bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
print('epe', d['epe_vs_oracle']['value'], 'hard', d['epe_hard_case']['value'], d['epe_hard_case']['worst_relative'], d['epe_hard_case']['within_1e-3_of_max(1,flow)'], d['epe_hard_case']['at_4_iterations']['worst_relative'])
print('host', d['host_ms_per_step'], d['host_cpu_ms_per_step'], d['graph_launch_calls'])
print('c2fp16', d['config2_fp16_mode']['value'], 'fp32', d['fp32_class_mode']['value'], 'single', d['single_clip']['value'], 'f2f', d['frames_to_flows_per_sec']['value'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
r=d['roofline']; print({k:v for k,v in r.items() if k not in ('per_kernel','method','traffic_note')})
for k in r['per_kernel']: print(k['kernel'], k['avg_us'], k['frac_hbm'], k['frac_mfma'], k['bound'], k['traffic'])
print(d['roofline_corr'])
d2=json.loads(open('gpurun_out/r06f/r06_bench_2ranks_shared.json').read().strip().splitlines()[-1])
print('2 ranks shared', d2['value'], d2['per_rank_ms_per_step'], d2['imbalance'], d2['config']['host_cores_per_rank'], d2['config']['device_pinning'], d2['host_ms_per_step'], d2['host_cpu_ms_per_step'])
d8=json.loads(open('gpurun_out/r06f/r06_bench_8ranks_shared.json').read().strip().splitlines()[-1])
print('8 ranks shared', d8['value'], d8['per_rank_ms_per_step'], d8['imbalance'], d8['config']['host_cores_per_rank'], d8['host_ms_per_step'], d8['host_cpu_ms_per_step'], d8['graph_launch_calls'])
PY
