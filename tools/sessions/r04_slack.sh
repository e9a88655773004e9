cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do timeout 900 python bench.py --no-cpu-baseline --no-kernel-breakdown 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"; done
