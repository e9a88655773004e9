#!/usr/bin/env python3
"""Correlation build + lookup, row-major fp16 volumes vs blocked fp16 volumes, at a BASELINE shape.
argv: workload (sintel|kitti|spring) [clips] [reps].  Prints one JSON line per variant (us, algorithmic GB/s by SURVEY 8d).
SF_CORR_TS=1 + a -DSF_CORR_TIMERS build prints the blocked build's phase timers."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops, synthetic as syn
from streamflow_amd.ops import Planes
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "sintel"
h, w, T = {"sintel": (55, 128, 4), "kitti": (47, 156, 2), "spring": (136, 240, 4)}[wl]
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"sintel": 8, "kitti": 8, "spring": 1}[wl]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
which = os.environ.get("SF_VARIANTS", "rows,blocked").split(",")
D, pairs, N = 256, T - 1, h * w
n = B * pairs
fm, _ = syn.make_features(1, B, T, h, w)
fm = fm.to(dev).contiguous()
g = torch.Generator().manual_seed(0)
from oracle import streamflow_oracle as orc   # (coords grid only: test infrastructure used by a tool, not the product)
coords = (orc.coords_grid(n, h, w) + torch.randn(n, 2, h, w, generator=g) * 3.0).to(dev).contiguous()
cp = Planes.of(coords)
cells = sum((h >> l) * (w >> l) for l in range(4))
b_bytes = n * (2.0 * N * D * 4 + 2.0 * N * cells)
l_bytes = n * (N * 4 * 100 * 2.0 + N * 2 * 4.0 + N * 324 * 4.0)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


res = {}
if "rows" in which:
    dims = [(h >> l, w >> l) for l in range(4)]
    strides = [B * N * a * b for a, b in dims]
    lv = [torch.empty(pairs * s, dtype=torch.float16, device=dev) for s in strides]
    ws = torch.empty(ops.corr_build_ws_bytes(B, pairs, D, h, w), dtype=torch.uint8, device=dev)
    out = torch.empty(n, 324, N, device=dev)
    op = Planes.of(out)
    op = Planes(op.base, op.off, op.img_stride, op.n_img, op.rows, op.P, shadow=ops.new_shadow(op, dev))
    tb = timed(lambda: ops.corr_build(fm.data_ptr(), fm.data_ptr() + 4 * D * N, T * D * N, D * N, lv, strides, B, pairs, D, h, w, ws=ws))
    tl = timed(lambda: ops.corr_lookup(lv, strides, cp, op, B, pairs, h, w))
    res["rows"] = (tb, tl)
    del lv, ws, out, op
    torch.cuda.empty_cache()
if "blocked" in which:
    vol = ops.new_blocked_volume(n, h, w, dev)
    ws = torch.empty(ops.corr_build_blocked_ws_bytes(n, D, h, w), dtype=torch.uint8, device=dev)
    ko = ops.new_shadow(Planes(torch.empty(1, device=dev), 0, 324 * N, n, 324, N), dev)
    build = lambda: ops.corr_build_blocked(fm.data_ptr(), fm.data_ptr() + 4 * D * N, T * D * N, D * N, vol, B, pairs, D, ws=ws)
    tb = timed(build)
    tl = timed(lambda: ops.corr_lookup_blocked(vol, cp, None, ko, B, pairs))
    res["blocked"] = (tb, tl)
    if os.environ.get("SF_CORR_TS"):
        ts = torch.zeros(131072 * 8, dtype=torch.int64, device=dev)
        os.environ["SF_CORR_TS_BUF"] = str(ts.data_ptr())
        build(); torch.cuda.synchronize()
        os.environ.pop("SF_CORR_TS_BUF")
        t = ts.view(-1, 8).cpu().double()
        t = t[t[:, 0] > 0]
        nt = t[:, 7].clamp(min=1)
        for name, v in (("wave life", t[:, 2] - t[:, 0]), ("patch loads", t[:, 1]), ("tiles/wave", t[:, 7]), ("k-loops", t[:, 5]), ("epilogues", t[:, 6]),
                        ("k-loop/tile", t[:, 5] / nt), ("epilogue/tile", t[:, 6] / nt)):
            print(f"# {name:13s} mean {v.mean().item():9.0f} median {v.median().item():9.0f} min {v.min().item():9.0f} max {v.max().item():9.0f}")
        dt_c, dt_r = t[:, 2] - t[:, 0], t[:, 4] - t[:, 3]
        print(f"# clock: {(dt_c.sum() / dt_r.sum()).item() * 100:.0f} MHz")
for k, (tb, tl) in res.items():
    it = 15
    frac = (b_bytes + it * l_bytes) / ((tb + it * tl) * 1e-6) / 8e12
    print(json.dumps({"workload": wl, "clips": B, "layout": k, "build_us": round(tb, 1), "build_gbps": round(b_bytes / tb / 1e3),
                      "lookup_us": round(tl, 1), "lookup_gbps": round(l_bytes / tl / 1e3),
                      "frac_of_8TBps_build_plus_15_lookups": round(frac, 4)}))
