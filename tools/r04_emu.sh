cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04e; mkdir -p $O
SF_EMU_X2_F16=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-kernel-breakdown > $O/emu.json 2>$O/emu.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04e/emu.json').read().strip().splitlines()[-1])
print(json.dumps(d['epe_vs_oracle'])[:500]); print(json.dumps(d['epe_hard_case'])[:300])
P
