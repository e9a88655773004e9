#!/bin/bash
# round-6: the Sintel PMC passes alone (HBM bytes, MFMA busy) -> gpurun_out/r06p/r06_traffic_sintel.json, r06_mfma_busy_sintel.json
cd "$GRAFT_REPO_ROOT"
SHA=$(python -c "import bench; print(bench.csrc_sha())" 2>/dev/null)
O=gpurun_out/r06p; mkdir -p $O
export TMPDIR=/tmp
P="$PWD"; cd /tmp
C="--steps 1 --warmup 0 --no-cpu-baseline --no-kernel-breakdown --no-graph"
export SF_ENGINE_OPTS="split_solo=0"
w=sintel
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/$O/pmc_${w}_fetch -o pmc -- python3 $P/bench.py $C --workload $w > $P/$O/pmc_${w}_fetch.log 2>&1; echo "pmc $w fetch rc $?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/$O/pmc_${w}_write -o pmc -- python3 $P/bench.py $C --workload $w > $P/$O/pmc_${w}_write.log 2>&1; echo "pmc $w write rc $?"
unset SF_ENGINE_OPTS
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/$O/pmc_sintel_mfma -o pmc -- python3 $P/bench.py $C > $P/$O/pmc_sintel_mfma.log 2>&1; echo "pmc mfma rc $?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $P/$O/pmc_sintel_lds -o pmc -- python3 $P/bench.py $C > $P/$O/pmc_sintel_lds.log 2>&1; echo "pmc lds rc $?"
cd $P
f=$(find $O/pmc_${w}_fetch -name "*counter_collection.csv" | head -1); g=$(find $O/pmc_${w}_write -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && [ -n "$g" ] && python tools/traffic_json.py $f $g _workload=$w _clips=8 _precision=f16x2 _corr_dtype=f16 _preset=config2_mixed _csrc_sha=$SHA > $O/r06_traffic_$w.json && python tools/pmc_summary.py $f $g > $O/r06_pmc_hbm_$w.md
f=$(find $O/pmc_sintel_mfma -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f > $O/r06_pmc_mfma_busy_sintel.md && python tools/pmc_mfma_json.py $f _workload=sintel _preset=config2_mixed _csrc_sha=$SHA > $O/r06_mfma_busy_sintel.json
f=$(find $O/pmc_sintel_lds -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f > $O/r06_pmc_lds_sintel.md
find $O/pmc_* -type f -name "*.csv" -delete 2>/dev/null; find $O -type f -name "*.db" -delete 2>/dev/null
python -c "
import json
t=json.load(open('$O/r06_traffic_sintel.json')); m=json.load(open('$O/r06_mfma_busy_sintel.json'))
for k in ('gemm','sk_tail','ffn_pair','gma_flash','corr_build','corr_lookup','dwconv15','temporal_block'):
    print(k, t.get(k), m.get(k))
"
