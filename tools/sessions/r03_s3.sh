#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
python tools/dbg_corrb.py 2>&1 | grep -v "bad cells 0 of" | tail -12
python -m pytest tests/test_gpu_corr_blocked.py -x -q 2>&1 | grep -E "Error|assert|first|level" | head -12
P="$PWD"; cd /tmp; export TMPDIR=/tmp
SF_VARIANTS=blocked timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/s3_trace -o t -- python3 $P/tools/corrb_bench.py sintel 8 5 > $P/$O/s3_trace.log 2>&1
cd $P; f=$(find $O/s3_trace -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
