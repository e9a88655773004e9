#!/bin/bash
# round-3 GPU session 2: where does the blocked build's time go (phase timers, ablations, knobs); PMC bytes of build + lookup
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03
mkdir -p $O
B=streamflow_amd/csrc/build
export SF_VARIANTS=blocked
echo "== base"; timeout 120 python tools/corrb_bench.py sintel 2>/dev/null | tail -1
echo "== timers"; SF_CORR_TS=1 SF_HIP_LIB=$B/variant_timers.so timeout 120 python tools/corrb_bench.py sintel 2>/dev/null | tail -3
for v in nost nok nt0 wg2; do echo "== $v"; SF_HIP_LIB=$B/variant_$v.so timeout 120 python tools/corrb_bench.py sintel 2>/dev/null | tail -1; done
echo "== 2 clips"; timeout 120 python tools/corrb_bench.py sintel 2 2>/dev/null | tail -1
echo "== 1 clip"; timeout 120 python tools/corrb_bench.py sintel 1 2>/dev/null | tail -1
P="$PWD"
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/$O/s2_pmc_$c -o pmc -- python3 $P/tools/corrb_bench.py sintel 8 2 > $P/$O/s2_pmc_$c.log 2>&1; echo "pmc $c rc $?"
done
cd $P
f=$(find $O/s2_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); g=$(find $O/s2_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $f $g > $O/s2_pmc_corr.md; cat $O/s2_pmc_corr.md
find $O/s2_pmc_* -type f ! -name "*counter_collection.csv" ! -name "*.log" -delete 2>/dev/null
