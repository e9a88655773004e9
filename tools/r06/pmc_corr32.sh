#!/bin/bash
# round-6: PMC passes of the fp32 corr-only step (BASELINE config 3): HBM bytes, MFMA busy, LDS -> gpurun_out/r06q/
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06q; mkdir -p $O
export TMPDIR=/tmp
P="$PWD"; cd /tmp
w=${1:-kitti}
C="--corr-only --workload $w --preset fp32_class --steps 1 --warmup 0 --no-cpu-baseline"
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES" "issue SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $P/$O/pmc_${w}_$name -o pmc -- python3 $P/bench.py $C > $P/$O/pmc_${w}_$name.log 2>&1 </dev/null; echo "pmc $w $name rc $?"
  f=$(find $P/$O/pmc_${w}_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $P/tools/pmc_summary.py $f > $P/$O/r06_pmc_corr32_${w}_$name.md </dev/null
done
find $P/$O/pmc_* -type f -name "*.csv" -delete 2>/dev/null; find $P/$O -type f -name "*.db" -delete 2>/dev/null
cat $P/$O/r06_pmc_corr32_${w}_*.md
