#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corr_blocked.py -x -q -k "not hard_case_sweep" 2>&1 | tail -4
echo "== flash A/B"; for i in 1 2; do timeout 120 python tools/flash_bench.py; SF_HIP_LIB=streamflow_amd/csrc/build/variant_prio.so timeout 120 python tools/flash_bench.py; done
N="--no-cpu-baseline --no-kernel-breakdown"
echo "== corr-only kitti fp32 (pitched / dense)"
timeout 300 python bench.py --corr-only --workload kitti --preset fp32_class > $O/bench_kitti_fp32_corr.json 2>/dev/null; python -c "
import json;d=json.loads(open('$O/bench_kitti_fp32_corr.json').read().strip().splitlines()[-1]);r=d['roofline'];print('pitched', r['frac'], r['build'], r['lookup'])"
timeout 300 python bench.py --corr-only --workload kitti --preset fp32_class --dense-volumes > $O/bench_kitti_fp32_corr_dense.json 2>/dev/null; python -c "
import json;d=json.loads(open('$O/bench_kitti_fp32_corr_dense.json').read().strip().splitlines()[-1]);r=d['roofline'];print('dense', r['frac'], r['build'], r['lookup'])"
echo "== setup overlap"; for o in 1 0; do SF_ENGINE_OPTS="setup_overlap=$o" timeout 300 python bench.py $N > $O/bench_overlap$o.json 2>/dev/null; python -c "
import json;d=json.loads(open('$O/bench_overlap$o.json').read().strip().splitlines()[-1]);print('setup_overlap=$o', d['value'], d['ms_per_step'])"; done
for q in 1 3; do timeout 600 python bench.py $N --preset fp32_class --gma flash --flash-qkp $q > $O/bench_fp32_flash$q.json 2>/dev/null; python -c "
import json;d=json.loads(open('$O/bench_fp32_flash$q.json').read().strip().splitlines()[-1]);print('fp32_class flash qkp $q', d['value'], d['ms_per_step'])"; done
timeout 2400 python tests/analysis/preset_select_v2.py > $O/r05_preset_select.jsonl 2> $O/sel.err; tail -3 $O/sel.err; tail -c 1500 $O/r05_preset_select.jsonl
