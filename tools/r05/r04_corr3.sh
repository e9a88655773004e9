#!/bin/bash
# BASELINE config 3 as worded: KITTI shape, fp32 level 0 + pyramid, build + lookups only: bench lines + rocprofv3 kernel stats + HBM counters
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04c; mkdir -p $O; P="$PWD"
C="--corr-only --workload kitti --preset fp32_class --clips 8"
timeout 300 python bench.py $C > $O/r04_bench_kitti_fp32_corr.json 2> $O/err.txt; tail -c 700 $O/r04_bench_kitti_fp32_corr.json; echo; tail -3 $O/err.txt
timeout 300 python bench.py --corr-only --workload kitti --clips 8 --preset config2_fp16 > $O/r04_bench_kitti_fp16_corr.json 2>/dev/null; tail -c 500 $O/r04_bench_kitti_fp16_corr.json; echo
timeout 300 python bench.py --corr-only --workload sintel --preset fp32_class --clips 8 > $O/r04_bench_sintel_fp32_corr.json 2>/dev/null; tail -c 500 $O/r04_bench_sintel_fp32_corr.json; echo
timeout 300 python bench.py --corr-only --workload kitti_w160 --preset fp32_class --clips 8 > $O/r04_bench_kittiw160_fp32_corr.json 2>/dev/null; tail -c 500 $O/r04_bench_kittiw160_fp32_corr.json; echo
timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -k "two_engines" 2>&1 | tail -5
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof -o prof -- python3 $P/bench.py $C --steps 5 --warmup 1 > $P/$O/prof.log 2>&1; echo "prof rc $?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/$O/pmc_fetch -o pmc -- python3 $P/bench.py $C --steps 1 --warmup 0 > $P/$O/pmc_fetch.log 2>&1; echo "fetch rc $?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/$O/pmc_write -o pmc -- python3 $P/bench.py $C --steps 1 --warmup 0 > $P/$O/pmc_write.log 2>&1; echo "write rc $?"
cd $P
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/stats_md.py $f "rocprofv3 --kernel-trace --stats: bench.py $C" > $O/r04_kernel_stats_kitti_fp32_corr.md
f=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); g=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && [ -n "$g" ] && python tools/pmc_summary.py $f $g > $O/r04_pmc_hbm_kitti_fp32_corr.md
find $O -type f -name "*.csv" -delete 2>/dev/null; find $O -type f -name "*.db" -delete 2>/dev/null
cat $O/r04_kernel_stats_kitti_fp32_corr.md | head -12; cat $O/r04_pmc_hbm_kitti_fp32_corr.md | head -20
