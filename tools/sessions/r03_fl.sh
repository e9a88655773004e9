#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
for lib in "" "$PWD/streamflow_amd/csrc/build/variant_flv0.so"; do
  SF_HIP_LIB=$lib python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib [$lib]', round(d['value'],1), d['kernels']['gma_flash']['ms_per_step'])"
done; done
