#!/usr/bin/env python3
"""Energy per launch of every distinct kernel launch of one hot-path step: the step is bound by the package power cap
(12.1), so what a kernel COSTS the step is its energy, not its time.  One eager forward of the bench configuration runs with a profiler hook
(ops.PROFILER) that, for the first occurrence of every distinct launch (name, shape, products), replays that launch alone in a loop for
`secs` seconds while a thread samples the GPU's hwmon power file; later occurrences are counted.  Prints per launch: count per step, us,
W while looping, mJ (total and above idle), GFLOP of issued MFMA work per J; and the per-family sums against the step's own energy.
usage: PYTHONPATH=. python tools/energy_table.py [--preset P] [--workload W] [--clips N] [--secs S]"""
import argparse, glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def rd(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


class Sampler(threading.Thread):
    def __init__(self, path):
        super().__init__(daemon=True)
        self.path, self.on, self.samples, self.stop = path, False, [], False

    def run(self):
        while not self.stop:
            if self.on:
                v = rd(self.path)
                if v is not None:
                    self.samples.append((time.perf_counter(), v / 1e6))
            time.sleep(0.01)


class EnergyProfiler:
    def __init__(self, sampler, secs):
        self.sampler, self.secs, self.recs, self.order = sampler, secs, {}, []

    def launch(self, name, flops, nbytes, fn, products=1.0):
        fn()
        key = (name, round(flops), round(nbytes), float(products))
        if key in self.recs:
            self.recs[key]["count"] += 1
            return
        if name.startswith("flow_update"):                       # in place (coords += delta): not replayed; ~5 us per launch
            self.recs[key] = {"name": name, "flops": flops, "bytes": nbytes, "products": products, "count": 1, "us": 5.0, "W": float("nan"), "loops": 0}
            return
        torch.cuda.synchronize()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        est = max(time.perf_counter() - t0, 5e-6)
        batch = max(1, min(2000, int(0.02 / est)))               # ~20 ms of launches between two synchronisations
        self.sampler.samples = []
        self.sampler.on = True
        n, t_start = 0, time.perf_counter()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        while time.perf_counter() - t_start < self.secs:
            for _ in range(batch):
                fn()
            n += batch
            torch.cuda.synchronize()
        e.record()
        torch.cuda.synchronize()
        self.sampler.on = False
        t_end = time.perf_counter()
        us = 1e3 * s.elapsed_time(e) / n
        # mean power over the second half of the loop (the sensor and the clock governor settle in the first)
        mid = t_start + 0.5 * (t_end - t_start)
        pw = [p for t, p in self.sampler.samples if t >= mid]
        watts = sum(pw) / len(pw) if pw else float("nan")
        self.recs[key] = {"name": name, "flops": flops, "bytes": nbytes, "products": products, "count": 1, "us": us, "W": watts, "loops": n}
        self.order.append(key)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default=None)
    ap.add_argument("--workload", default="sintel")
    ap.add_argument("--clips", type=int, default=8)
    ap.add_argument("--secs", type=float, default=0.6)
    ap.add_argument("--gma", default=None, help="gma_mode override (flash / stored / hybrid / matrix)")
    args = ap.parse_args()
    import bench
    from streamflow_amd import ops, presets, synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    H, W, T, iters = bench.WORKLOADS[args.workload]
    h, w, B = H // 8, W // 8, args.clips
    dev = torch.device("cuda:0")
    hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    pfiles = [d + ("/power1_average" if os.path.exists(d + "/power1_average") else "/power1_input") for d in hw]
    if not pfiles:
        raise SystemExit("no hwmon power files on this box")
    # which GPU is ours: the one whose power rises under a second of matrix products
    idle = [rd(p) or 0 for p in pfiles]
    x = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
    t0 = time.time()
    peak = list(idle)
    while time.time() - t0 < 1.5:
        for _ in range(20):
            y = x @ x
        torch.cuda.synchronize()
        peak = [max(a, rd(p) or 0) for a, p in zip(peak, pfiles)]
    del x, y
    g = max(range(len(pfiles)), key=lambda i: peak[i] - idle[i])
    time.sleep(2.0)
    p_idle = (rd(pfiles[g]) or 0) / 1e6
    print(f"power file {pfiles[g]}; idle {p_idle:.0f} W; cap {(rd(hw[g] + '/power1_cap') or 0) / 1e6:.0f} W", flush=True)
    sampler = Sampler(pfiles[g])
    sampler.start()
    cfg = presets.engine_kwargs(args.preset or presets.BENCH_PRESET)
    if args.gma:
        cfg["gma_mode"] = args.gma
    params = syn.make_params(0, T)
    fmaps, cnets = syn.make_features(1000, B, T, h, w)
    fmaps, cnets = fmaps.to(dev), cnets.to(dev)
    eng = HotPathEngine(params, device=dev, T=T, use_graph=False, **cfg)
    eng.parallel_branches = False
    eng.forward(fmaps, cnets, iters=iters)                      # buffers, packs
    torch.cuda.synchronize()
    # the step's own power and time (graph replay, the shipped schedule)
    eng2 = HotPathEngine(params, device=dev, T=T, use_graph=True, **cfg)
    for _ in range(3):
        eng2.forward(fmaps, cnets, iters=iters)
    torch.cuda.synchronize()
    sampler.samples, sampler.on = [], True
    t0 = time.perf_counter()
    n_steps = 0
    while time.perf_counter() - t0 < 4.0:
        eng2.forward(fmaps, cnets, iters=iters)
        n_steps += 1
        if n_steps % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    sampler.on = False
    pw = [p for t, p in sampler.samples if t >= t0 + 0.5 * (t1 - t0)]
    step_ms, step_w = 1e3 * (t1 - t0) / n_steps, sum(pw) / len(pw)
    print(f"step (graph replay): {step_ms:.2f} ms at {step_w:.0f} W = {step_ms * step_w / 1e3:.1f} J ({step_ms * (step_w - p_idle) / 1e3:.1f} J above idle)", flush=True)
    del eng2
    ops.PROFILE_SHAPES = True
    prof = EnergyProfiler(sampler, args.secs)
    ops.PROFILER = prof
    eng.forward(fmaps, cnets, iters=iters)
    torch.cuda.synchronize()
    ops.PROFILER = None
    sampler.stop = True
    print("| launch | per step | us | W alone | mJ | mJ above idle | J per step | issued MFMA GFLOP per J above idle |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    fam, tot_j, tot_jd, tot_ms = {}, 0.0, 0.0, 0.0
    for key in prof.order:
        r = prof.recs[key]
        if r["W"] != r["W"]:
            continue
        mj, mjd = r["us"] * r["W"] / 1e3, r["us"] * (r["W"] - p_idle) / 1e3
        j = r["count"] * mj / 1e3
        issued = r["flops"] * r["products"]
        eff = issued / 1e9 / (mjd / 1e3) if (issued > 0 and mjd > 0) else 0.0
        print(f"| {r['name']} | {r['count']} | {r['us']:.1f} | {r['W']:.0f} | {mj:.1f} | {mjd:.1f} | {j:.2f} | {eff:.0f} |")
        f = r["name"].split(" ")[0]
        d = fam.setdefault(f, [0, 0.0, 0.0, 0.0, 0.0])
        d[0] += r["count"]; d[1] += r["count"] * r["us"] / 1e3; d[2] += j; d[3] += r["count"] * mjd / 1e3; d[4] += r["count"] * issued
        tot_j += j; tot_jd += r["count"] * mjd / 1e3; tot_ms += r["count"] * r["us"] / 1e3
    print()
    print("| family | launches per step | ms per step (alone, serial) | J per step | J above idle | share of the J above idle | issued MFMA TFLOP |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for f, d in sorted(fam.items(), key=lambda kv: -kv[1][3]):
        print(f"| {f} | {d[0]} | {d[1]:.2f} | {d[2]:.2f} | {d[3]:.2f} | {100 * d[3] / tot_jd:.1f} % | {d[4] / 1e12:.2f} |")
    print(f"| all | | {tot_ms:.2f} | {tot_j:.1f} | {tot_jd:.1f} | | |")
    print(f"step as shipped: {step_ms:.2f} ms, {step_ms * step_w / 1e3:.1f} J, {step_ms * (step_w - p_idle) / 1e3:.1f} J above idle")


if __name__ == "__main__":
    main()
