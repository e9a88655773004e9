#!/bin/bash
# round-2 GPU session 1: full GPU test suite, default bench, HBM-sensitivity experiment, MFMA-busy / LDS counters
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
( timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest_s1.log 2>&1; echo "pytest rc $?" >> $O/pytest_s1.log )
tail -3 $O/pytest_s1.log
timeout 600 python bench.py > $O/bench_s1.json 2> $O/bench_s1.err; echo "bench rc $?"
tail -c 600 $O/bench_s1.json
python tools/gemm_alias.py 960 640 gelu 24 > $O/alias.log 2>&1
python tools/gemm_alias.py 640 960 none 24 >> $O/alias.log 2>&1
python tools/gemm_alias.py 384 256 gelu 24 >> $O/alias.log 2>&1
cat $O/alias.log
for c in 2 4; do timeout 300 python bench.py --clips $c --no-cpu-baseline --no-kernel-breakdown > $O/bench_clips$c.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_clips$c.json'));print('clips',$c,d['value'],d['ms_per_step'])"; done
rocprofv3 -L > $O/counters_list.txt 2>&1
P="$PWD"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $P/$O/pmcA -o pmcA --output-format csv -- python3 $P/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-breakdown --no-graph > $P/$O/pmcA.log 2>&1; echo "pmcA rc $?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum -d $P/$O/pmcB -o pmcB --output-format csv -- python3 $P/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-breakdown --no-graph > $P/$O/pmcB.log 2>&1; echo "pmcB rc $?"
cd $P
find $O/pmcA $O/pmcB -name "*counter_collection.csv" | head
for d in pmcA pmcB; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f > $O/${d}_summary.md; done
# keep the merged output small: drop raw csv/db beyond the summaries
find $O/pmcA $O/pmcB -type f ! -name "*counter_collection.csv" -delete 2>/dev/null
du -sh $O
