"""A/B of the f16x3 GEMM-to-GEMM hand-over: fp32 planes vs split k-octet planes (SF_LAYOUT_SPLIT_KOCT / c_f16 = 4), per layer shape,
24 images x 7040 pixels.  usage: PYTHONPATH=. python tools/gemm_split_koct_bench.py"""
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear

dev = torch.device("cuda:0")
n, P = 24, 7040
ops.set_precision("f16x3")


def split_planes(rows):
    r8 = (rows + 7) // 8 * 8
    return Planes(torch.zeros(n * r8 * P, device=dev), 0, 2 * r8 * P, n, rows, P, f16=True, koct=True, split=True)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


g = torch.Generator().manual_seed(0)
for C, H in ((256, 384), (324, 486), (384, 576), (640, 960), (128, 192)):
    W1 = PackedLinear((torch.randn(H, C, generator=g) / C ** 0.5).reshape(H, C, 1, 1), torch.zeros(H), dev)
    W2 = PackedLinear((torch.randn(C, H, generator=g) / H ** 0.5).reshape(C, H, 1, 1), torch.zeros(C), dev)
    X = Planes.of(torch.randn(n, C, P, generator=g).to(dev))
    h32, hs = Planes.of(torch.zeros(n, H, P, device=dev)), split_planes(H)
    Y = Planes.of(torch.zeros(n, C, P, device=dev))
    t = [timeit(lambda: ops.gemm(W1, X, h32, ops.EPI_GELU)), timeit(lambda: ops.gemm(W1, X, hs, ops.EPI_GELU)),
         timeit(lambda: ops.gemm(W2, h32, Y, ops.EPI_NONE)), timeit(lambda: ops.gemm(W2, hs, Y, ops.EPI_NONE))]
    fl = 2.0 * 3 * C * H * n * P
    print(f"C{C} H{H}: producer fp32 out {t[0]:.1f} us, split out {t[1]:.1f} us | consumer fp32 B {t[2]:.1f} us ({fl / t[2] / 1e6:.0f} TF issued), "
          f"split B {t[3]:.1f} us ({fl / t[3] / 1e6:.0f} TF)", flush=True)
