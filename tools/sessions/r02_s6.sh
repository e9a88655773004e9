#!/bin/bash
# round-2 GPU session 6: default bench, rocprofv3 kernel stats + PMC passes (sintel and kitti), spring
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
: # (full GPU suite run separately)
:
timeout 900 python bench.py > $O/bench_s6.json 2> $O/bench_s6.err; echo "bench rc $?"
python - <<PY
import json
d=json.load(open('$O/bench_s6.json'))
print('default', round(d['value'],1), round(d['ms_per_step'],2), d['dtype'], d['roofline'], d.get('roofline_corr'), d.get('epe_vs_oracle'), d.get('fp32_class_mode'), d.get('single_clip'))
PY
P="$PWD"
cd /tmp
B="--steps 3 --warmup 1 --no-cpu-baseline --no-kernel-breakdown"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof_sintel -o prof -- python3 $P/bench.py $B > $P/$O/prof_sintel.log 2>&1; echo "prof sintel rc $?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof_sintel_serial -o prof -- python3 $P/bench.py $B --no-graph --serial-branches > $P/$O/prof_sintel_serial.log 2>&1; echo "prof sintel serial rc $?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof_kitti -o prof -- python3 $P/bench.py $B --workload kitti > $P/$O/prof_kitti.log 2>&1; echo "prof kitti rc $?"
C="--steps 1 --warmup 0 --no-cpu-baseline --no-kernel-breakdown --no-graph"
# (HBM-traffic passes on the ONE-chain schedule: bench.py prices its instrumented, unsplit step with these per-launch bytes)
export SF_SPLIT_SOLO=0
for w in sintel kitti; do
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/$O/pmc_${w}_fetch -o pmc -- python3 $P/bench.py $C --workload $w > $P/$O/pmc_${w}_fetch.log 2>&1; echo "pmc $w fetch rc $?"
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/$O/pmc_${w}_write -o pmc -- python3 $P/bench.py $C --workload $w > $P/$O/pmc_${w}_write.log 2>&1; echo "pmc $w write rc $?"
done
unset SF_SPLIT_SOLO
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/$O/pmc_sintel_mfma -o pmc -- python3 $P/bench.py $C > $P/$O/pmc_sintel_mfma.log 2>&1; echo "pmc mfma rc $?"
cd $P
# keep only stats / counter csv (small)
find $O/prof_sintel $O/prof_sintel_serial $O/prof_kitti -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
find $O/pmc_* -type f ! -name "*counter_collection.csv" ! -name "*.log" -delete 2>/dev/null
for w in sintel kitti; do
  f=$(find $O/pmc_${w}_fetch -name "*counter_collection.csv" | head -1); g=$(find $O/pmc_${w}_write -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && [ -n "$g" ] && python tools/traffic_json.py $f $g _workload=$w _clips=8 _precision=f16x2 _corr_dtype=f16 _preset=config2_fp16 > $O/traffic_$w.json && python tools/pmc_summary.py $f $g > $O/pmc_${w}_hbm.md
done
f=$(find $O/pmc_sintel_mfma -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f > $O/pmc_sintel_mfma.md
find $O/pmc_* -name "*counter_collection.csv" -size +8M -delete 2>/dev/null
timeout 400 python bench.py --workload kitti --no-cpu-baseline > $O/bench_s6_kitti.json 2>/dev/null
timeout 400 python bench.py --workload spring --clips 1 --steps 5 --no-cpu-baseline > $O/bench_s6_spring.json 2>/dev/null
timeout 400 python bench.py --preset fp32_class --no-cpu-baseline > $O/bench_s6_fp32class.json 2>/dev/null
timeout 400 python bench.py --clips 1 --no-cpu-baseline > $O/bench_s6_clip1.json 2>/dev/null
python - <<PY
import json
for n in ('kitti','spring','fp32class','clip1'):
    try:
        d=json.load(open('$O/bench_s6_%s.json'%n)); print(n, round(d['value'],1), round(d['ms_per_step'],2), d.get('roofline_corr',{}).get('frac'))
    except Exception as e: print(n,'failed',e)
PY
du -sh $O; ls $O | head -60
