cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_gemm_bstat.py tests/test_gpu_parity.py -q -x -k "bstat or skblock or pw_fold or handover" 2>&1 | tail -4
echo "== transpose"; timeout 300 python tools/pw_b_format.py 2>&1 | tail -5
echo "== eight 2-byte loads"; SF_HIP_LIB=streamflow_amd/csrc/build/variant_rowsold.so timeout 300 python tools/pw_b_format.py 2>&1 | tail -5
