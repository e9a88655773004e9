#!/bin/bash
# power / clock samples while (a) the two-product K = 640 GEMM loops, (b) the bench step loops
mkdir -p gpurun_out/r05t
python -m streamflow_amd.build > /dev/null 2>&1
smi() { for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showtemp --showperflevel 2>&1 | grep -i -E "power|sclk|mclk|fclk|Temp|perf" | tr '\n' ';'; echo; sleep 1.5; done; }
{
echo "== idle"; rocm-smi --showpower --showclocks --showtemp --showmaxpower 2>&1 | grep -v "^=\|^$" | head -30
echo "== two-product GEMM loop (M960 K640)"
( SF_SHAPES=960x640 SF_LOOPS=1 timeout 60 python - <<'P'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear
dev = torch.device("cuda:0"); ops.set_precision("f16x2")
M, K, n, P = 960, 640, 24, 7040
W = PackedLinear(torch.randn(M, K, 1, 1) / K ** 0.5, torch.randn(M) * 0.1, dev)
X = Planes(torch.zeros(n * K * P // 2, device=dev), 0, K * P, n, K, P, f16=True, koct=True)
ops.pack_koct(Planes.of(torch.randn(n, K, P, device=dev)), X)
Y = Planes(torch.zeros(n * M * P // 2, device=dev), 0, M * P, n, M, P, f16=True, koct=True)
for single in (False, True):
    W.single = single
    for zero in (False, True):
        if zero: X.t.zero_()
        t0 = time.time(); its = 0
        while time.time() - t0 < 10:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(200): ops.gemm(W, X, Y, ops.EPI_GELU)
            e.record(); torch.cuda.synchronize(); its += 1
            last = s.elapsed_time(e) * 5
        print(f"single={single} zero_inputs={zero}: {last:.1f} us per launch (after {its} x 200 launches)", flush=True)
    ops.pack_koct(Planes.of(torch.randn(n, K, P, device=dev)), X)
P
) &
pid=$!
sleep 14; smi; sleep 2; smi; sleep 4; smi; sleep 2; smi
wait $pid
echo "== bench step loop"
( python bench.py --steps 150 --warmup 3 --no-cpu-baseline --no-kernel-breakdown 2>/dev/null | cut -c1-150 ) &
pid=$!
sleep 45; smi
wait $pid
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05t/power.txt
