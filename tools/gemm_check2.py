import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear
dev = torch.device("cuda:0")
ops.set_precision("f16x3")
M, K, P, n = 64, 32, 128, 1
W = PackedLinear(torch.zeros(M, K), torch.zeros(M), dev)
x = torch.zeros(n, K, P, device=dev)
r = (torch.arange(P, device=dev).float()[None, None, :] + 1000 * torch.arange(M, device=dev).float()[None, :, None]).contiguous()
y = torch.empty(n, M, P, device=dev)
ops.gemm(W, Planes.of(x), Planes.of(y), ops.EPI_RES, R=Planes.of(r))
print(y[0, 0, :16].tolist())
print(y[0, 1, :16].tolist())
print(y[0, 9, 32:48].tolist())
