#!/bin/bash
# rocprofv3 kernel stats of the Twins_CSC encoder alone (fnet + cnet, 8 clips, both precision classes)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03p; mkdir -p $O
export TMPDIR=/tmp
P="$PWD"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof_encoder -o prof -- python3 $P/tools/encoder_bench.py 8 > $P/$O/prof_encoder.log 2>&1; echo "prof encoder rc $?"
cd $P
f=$(find $O/prof_encoder -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/stats_md.py $f "rocprofv3 --kernel-trace --stats: tools/encoder_bench.py 8 (Twins_CSC fnet + cnet, 8 clips at 440x1024; f16x3 then f16x2 passes)" > $O/r03_kernel_stats_encoder.md
find $O/prof_encoder -type f -name "*.csv" -delete 2>/dev/null; find $O -type f -name "*.db" -delete 2>/dev/null
tail -2 $O/prof_encoder.log | cut -c1-400; head -16 $O/r03_kernel_stats_encoder.md
