cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04e
for o in "split_solo=2" "split_solo=4" "split_solo=0" "parallel_branches=0"; do
SF_ENGINE_OPTS="$o" timeout 900 python bench.py --no-cpu-baseline --no-kernel-breakdown > gpurun_out/r04e/bench_ss.json 2>/dev/null; python - "$o" <<'P'
import json,sys
d=json.loads(open('gpurun_out/r04e/bench_ss.json').read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value'],1), round(d['ms_per_step'],2))
P
done
