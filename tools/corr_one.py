#!/usr/bin/env python3
"""Correlation-volume build alone at the Sintel shape (timing / rocprofv3 runs).  argv: reps [h w]
SF_CLIPS=8 sets the clips, SF_CORR_DTYPE=f16|f32 the volume format; SF_CORR_TS=1 prints per-workgroup phase timers of
the fp16 build (needs tools/build_variant.sh timers corr.hip -DSF_CORR_TIMERS and SF_HIP_LIB=.../variant_timers.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops, synthetic as syn
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (55, 128)
B, T, D = int(os.environ.get("SF_CLIPS", "1")), 4, 256
f16 = os.environ.get("SF_CORR_DTYPE", "f32") == "f16"
fm, _ = syn.make_features(1, B, T, h, w)
fm = fm.to(dev).contiguous()
P = h * w
dims = [(h >> l, w >> l) for l in range(4)]
strides = [B * P * a * b for a, b in dims]
lv = [torch.empty((T - 1) * s, dtype=torch.float16 if f16 else torch.float32, device=dev) for s in strides]
ws = torch.empty(ops.corr_build_ws_bytes(B, T - 1, D, h, w), dtype=torch.uint8, device=dev)
def run():
    ops.corr_build(fm.data_ptr(), fm.data_ptr() + 4 * D * P, T * D * P, D * P, lv, strides, B, T - 1, D, h, w, ws=ws)
for _ in range(3):
    run()
torch.cuda.synchronize()
if os.environ.get("SF_CORR_TS"):
    ts = torch.zeros(65536 * 8, dtype=torch.int64, device=dev)
    os.environ["SF_CORR_TS_BUF"] = str(ts.data_ptr())
    run(); torch.cuda.synchronize()
    os.environ.pop("SF_CORR_TS_BUF")
    t = ts.view(-1, 8).cpu().double()
    t = t[t[:, 0] > 0]
    print("workgroups", t.shape[0])
    for name, v in (("k-loop", t[:, 1] - t[:, 0]), ("epilogue", t[:, 2] - t[:, 1])):
        print(f"{name:9s} mean {v.mean().item():9.0f} median {v.median().item():9.0f} min {v.min().item():9.0f} max {v.max().item():9.0f} cycles")
    dt_c, dt_r = t[:, 2] - t[:, 0], t[:, 4] - t[:, 3]
    print(f"clock: {(dt_c.sum() / dt_r.sum()).item() * 100:.0f} MHz")
    span = (t[:, 4].max() - t[:, 3].min()).item() / 100.0
    print(f"kernel span {span:.1f} us; resident workgroups on average {dt_r.sum().item() / 100.0 / span:.1f} (max 768)")
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    run()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / reps
cells = sum(a * b for a, b in dims)
nbytes = B * (T - 1) * (2.0 * P * D * 4 + (2.0 if f16 else 4.0) * P * cells)
print(f"corr_build {us:.1f} us  {nbytes / us / 1e3:.0f} GB/s algorithmic  {2.0 * P * P * D * B * (T - 1) / us / 1e6:.1f} TF")
