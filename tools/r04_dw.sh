cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "gma_flash or flash" 2>&1 | tail -3
for i in 1 2; do
echo "== prefetch"; timeout 300 python tools/flash_bench.py; timeout 300 python tools/flash_bench.py 3 7040
echo "== plain"; SF_HIP_LIB=streamflow_amd/csrc/build/variant_nopf.so timeout 300 python tools/flash_bench.py; SF_HIP_LIB=streamflow_amd/csrc/build/variant_nopf.so timeout 300 python tools/flash_bench.py 3 7040
done
