cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; P="$PWD"; O=gpurun_out/r05p; mkdir -p $O; cd /tmp
C="--steps 1 --warmup 0 --no-cpu-baseline --no-kernel-breakdown --no-graph"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/$O/pmc_sintel_lds -o pmc -- python3 $P/bench.py $C > $P/$O/pmc_sintel_lds.log 2>&1; echo "rc $?"
cd $P
f=$(find $O/pmc_sintel_lds -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f > $O/r05_pmc_lds_sintel.md; head -40 $O/r05_pmc_lds_sintel.md; tail -3 $O/pmc_sintel_lds.log
find $O/pmc_sintel_lds -type f -name "*.csv" -delete 2>/dev/null; find $O -type f -name "*.db" -delete 2>/dev/null
