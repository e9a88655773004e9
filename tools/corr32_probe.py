"""Per-step GPU and host times of the blocked fp32 corr-only step (debug aid: an intermittent 2x in bench.py's ms_per_step)."""
import sys, time
import torch
from streamflow_amd import ops, synthetic as syn
from streamflow_amd.ops import Planes

dev = torch.device("cuda:0")
H, W, T, B, D, iters = 376, 1248, 2, 8, 256, 15
h, w, pairs = H // 8, W // 8, T - 1
N, n = h * w, B * pairs
fmaps = syn.make_features(1000, B, T, h, w)[0].to(dev)
vol = ops.new_blocked_volume(n, h, w, dev, f32=True)
ws = torch.empty(max(ops.corr_build_blocked_ws_bytes(n, D, h, w, True), 16), dtype=torch.uint8, device=dev)
g = torch.Generator().manual_seed(7)
ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
grid = torch.stack([xs, ys])[None]
coords = [Planes.of((grid + 4.0 * torch.randn(n, 2, h, w, generator=g)).reshape(n, 2, N).contiguous().to(dev)) for _ in range(iters)]
out = Planes.of(torch.empty(n, 324, N, device=dev))


def step():
    ops.corr_build_blocked(fmaps.data_ptr(), fmaps.data_ptr() + 4 * D * N, T * D * N, D * N, vol, B, pairs, D, ws=ws)
    for c in coords:
        ops.corr_lookup_blocked(vol, c, out, None, B, pairs)


step()
torch.cuda.synchronize()
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(24)]
    host = []
    t0 = time.perf_counter()
    for i in range(23):
        ev[i].record()
        t = time.perf_counter()
        step()
        host.append(1e3 * (time.perf_counter() - t))
    ev[23].record()
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - t0)
    gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(23)]
    print(f"rep {rep}: wall {wall:.2f} ms; gpu per step min {min(gpu):.2f} max {max(gpu):.2f} mean {sum(gpu) / 23:.2f}; "
          f"host per step min {min(host):.2f} max {max(host):.2f} mean {sum(host) / 23:.2f}", flush=True)
