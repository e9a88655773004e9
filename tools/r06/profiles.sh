#!/bin/bash
# round-6 profiles: rocprofv3 kernel stats (graph + serial, kitti, spring), PMC HBM bytes (sintel, kitti, spring), MFMA busy
cd "$GRAFT_REPO_ROOT"
SHA=$(python -c "import bench; print(bench.csrc_sha())" 2>/dev/null)
echo "csrc sha $SHA"
O=gpurun_out/r06p; mkdir -p $O
export TMPDIR=/tmp
P="$PWD"; cd /tmp
B="--steps 3 --warmup 1 --no-cpu-baseline --no-kernel-breakdown"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof_sintel -o prof -- python3 $P/bench.py $B > $P/$O/prof_sintel.log 2>&1; echo "prof sintel rc $?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof_sintel_serial -o prof -- python3 $P/bench.py $B --no-graph --serial-branches > $P/$O/prof_sintel_serial.log 2>&1; echo "prof sintel serial rc $?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof_kitti -o prof -- python3 $P/bench.py $B --workload kitti > $P/$O/prof_kitti.log 2>&1; echo "prof kitti rc $?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof_spring -o prof -- python3 $P/bench.py $B --workload spring --clips 1 > $P/$O/prof_spring.log 2>&1; echo "prof spring rc $?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof_fp32class -o prof -- python3 $P/bench.py $B --preset fp32_class > $P/$O/prof_fp32class.log 2>&1; echo "prof fp32_class rc $?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof_kitti_fp32_corr -o prof -- python3 $P/bench.py --corr-only --workload kitti --preset fp32_class --no-cpu-baseline > $P/$O/prof_kitti_fp32_corr.log 2>&1; echo "prof kitti fp32 corr rc $?"
C="--steps 1 --warmup 0 --no-cpu-baseline --no-kernel-breakdown --no-graph"
export SF_ENGINE_OPTS="split_solo=0"
for w in sintel kitti spring; do
  X=""; [ $w = spring ] && X="--clips 1"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/$O/pmc_${w}_fetch -o pmc -- python3 $P/bench.py $C --workload $w $X > $P/$O/pmc_${w}_fetch.log 2>&1; echo "pmc $w fetch rc $?"
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/$O/pmc_${w}_write -o pmc -- python3 $P/bench.py $C --workload $w $X > $P/$O/pmc_${w}_write.log 2>&1; echo "pmc $w write rc $?"
done
unset SF_ENGINE_OPTS
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/$O/pmc_sintel_mfma -o pmc -- python3 $P/bench.py $C > $P/$O/pmc_sintel_mfma.log 2>&1; echo "pmc mfma rc $?"
cd $P
for w in sintel sintel_serial kitti spring; do f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/stats_md.py $f "rocprofv3 --kernel-trace --stats: bench.py $w (config2_mixed)" > $O/r06_kernel_stats_$w.md; done
f=$(find $O/prof_fp32class -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/stats_md.py $f "rocprofv3 --kernel-trace --stats: bench.py --preset fp32_class (sintel)" > $O/r06_kernel_stats_fp32class.md
f=$(find $O/prof_kitti_fp32_corr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/stats_md.py $f "rocprofv3 --kernel-trace --stats: bench.py --corr-only --workload kitti --preset fp32_class (BASELINE config 3)" > $O/r06_kernel_stats_kitti_fp32_corr.md
for w in sintel kitti spring; do
  f=$(find $O/pmc_${w}_fetch -name "*counter_collection.csv" | head -1); g=$(find $O/pmc_${w}_write -name "*counter_collection.csv" | head -1)
  c=8; [ $w = spring ] && c=1
  [ -n "$f" ] && [ -n "$g" ] && python tools/traffic_json.py $f $g _workload=$w _clips=$c _precision=f16x2 _corr_dtype=f16 _preset=config2_mixed _csrc_sha=$SHA > $O/r06_traffic_$w.json && python tools/pmc_summary.py $f $g > $O/r06_pmc_hbm_$w.md
done
f=$(find $O/pmc_sintel_mfma -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f > $O/r06_pmc_mfma_busy_sintel.md && python tools/pmc_mfma_json.py $f _workload=sintel _preset=config2_mixed _csrc_sha=$SHA > $O/r06_mfma_busy_sintel.json
find $O/prof_* $O/pmc_* -type f -name "*.csv" -delete 2>/dev/null; find $O -type f -name "*.db" -delete 2>/dev/null
ls $O; cat $O/r06_mfma_busy_sintel.json; head -14 $O/r06_kernel_stats_sintel_serial.md
