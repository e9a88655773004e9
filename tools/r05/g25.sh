#!/bin/bash
mkdir -p gpurun_out/r05t
{
b() { python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-kernel-breakdown 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for o in "" "split_solo=0" "split_solo=4" "parallel_branches=0" "setup_overlap=0" "project_v=0"; do echo "opts [$o]:"; SF_ENGINE_OPTS=$o b; done
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
} 2>&1 | tee gpurun_out/r05t/sched_ab.txt
