#!/bin/bash
# What clock / power does the GPU sustain under a long GEMM loop?  (run on the GPU box)
rocm-smi --showclocks --showpower --showmaxpower --showperflevel 2>&1 | grep -v "^$" | grep -v "====" | head -30
python3 - <<'PY' &
import sys, os, torch, time
sys.path.insert(0, os.getcwd())
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear
dev = torch.device("cuda:0")
W = PackedLinear(torch.randn(960, 640) / 25, torch.randn(960) * 0.1, dev)
X = Planes.of(torch.randn(3, 640, 7040, device=dev)); Y = Planes.of(torch.empty(3, 960, 7040, device=dev))
t0 = time.time()
while time.time() - t0 < 20:
    for _ in range(200): ops.gemm(W, X, Y, ops.EPI_GELU)
    torch.cuda.synchronize()
PY
sleep 12
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power|fclk|mclk|socclk" | head -8; echo; sleep 2; done
wait
