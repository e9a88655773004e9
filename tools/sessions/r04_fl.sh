cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
echo "== V1=1 (ship)"; timeout 300 python tools/flash_bench.py
echo "== V1=0"; SF_HIP_LIB=streamflow_amd/csrc/build/variant_fv0.so timeout 300 python tools/flash_bench.py
done
