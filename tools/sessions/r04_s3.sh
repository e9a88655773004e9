#!/bin/bash
# round-4 session 3: what bounds the weight stream of the activation-stationary GEMM?  ring depth (latency) vs replicas (L2 hot spots)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
B=streamflow_amd/csrc/build
export SF_SHAPES=960x640,640x640,384x256,128x128 SF_ALGO=2
for rep in 1 2; do
for single in 1 0; do
  echo "== rep $rep single=$single base";  SF_SINGLE=$single timeout 200 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M"
  echo "== rep $rep single=$single ring4"; SF_SINGLE=$single SF_HIP_LIB=$B/variant_ring4.so timeout 200 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M"
  echo "== rep $rep single=$single repl2"; SF_SINGLE=$single SF_REPL=2 SF_HIP_LIB=$B/variant_repl2.so timeout 200 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M"
  echo "== rep $rep single=$single repl3"; SF_SINGLE=$single SF_REPL=3 SF_HIP_LIB=$B/variant_repl3.so timeout 200 python tools/gemm_koct_bench.py koct 2>&1 | grep "^M"
done; done > $O/s3_variants.log 2>&1
cat $O/s3_variants.log
