import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from dataclasses import replace
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear, PackedPair
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_ffn_pair import _koct, _weff, _layers
dev = torch.device("cuda:0")
K1, H, M2, P, pm, n = 128, 192, 64, 64, (1, 1), 3
A1, A2, pair, b1, b2, g = _layers(K1, H, M2, 1, dev, pm)
x = torch.randn(n, K1, P, generator=g)
X = _koct(x, dev, ops)
cx = ops.Ctx(precision=ops.PRECISION_F16X2)
x16 = x.half().double()
hid = F.gelu(torch.einsum("hk,nkp->nhp", _weff(A1, True), x16) + b1.double()[None, :, None]).half().double()
ref = torch.einsum("mh,nhp->nmp", _weff(A2, True), hid) + b2.double()[None, :, None]
Mo = 64
Yk = Planes(torch.full((n * Mo * P // 2 + 8,), float("nan"), device=dev), 0, Mo * P, n, M2, P, f16=True, koct=True)
ops.ffn_pair(pair, X, Yk, 0, cx=cx)
gk = Yk.tensor().double().cpu()
print("koct-only err", (gk - ref).abs().max().item())
y32 = torch.full((n, M2, P), float("nan"), device=dev)
Y = Planes.of(y32)
ops.ffn_pair(pair, X, Y, 0, cx=cx)
torch.cuda.synchronize()
e = (y32.double().cpu() - ref).abs()
print("fp32-only err", e.max().item(), "nan", int(torch.isnan(y32).sum()))
bad = (e > 1e-2).nonzero()
print("bad count", len(bad), "of", e.numel(), "first", bad[:12].tolist())
print("rows bad:", sorted(set(bad[:, 1].tolist()))[:40])
print("px bad:", sorted(set(bad[:, 2].tolist()))[:70])
# does y32 equal ref at permuted rows?
d = y32.double().cpu()
for r in (0, 1, 4, 5, 16, 17):
    match = [(rr) for rr in range(M2) if (d[0, r] - ref[0, rr]).abs().max() < 1e-2]
    print("row", r, "matches ref rows", match)
