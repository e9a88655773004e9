#!/usr/bin/env python3
"""Socket power and clocks of the GPU while bench.py replays steps (evidence for 'the step is power-bound'):
samples the amdgpu hwmon files (power1_average / power1_input, power1_cap, freq1_input) every 20 ms from THIS process (which never
touches the GPU runtime) while a child runs `bench.py --steps N`.   usage: power_log.py [bench args ...]"""
import glob, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hwmons():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        out.append(d)
    return out


def rd(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


hw = hwmons()
print("hwmon dirs:", hw)
for d in hw:
    print(d, {os.path.basename(p): rd(p) for p in sorted(glob.glob(d + "/power1_*") + glob.glob(d + "/freq*_input"))})
if not hw:
    r = subprocess.run(["rocm-smi", "--showpower", "--showmaxpower", "--showclocks"], capture_output=True, text=True)
    print(r.stdout[-2000:], r.stderr[-500:])
    sys.exit(0)
pfiles = [d + ("/power1_average" if os.path.exists(d + "/power1_average") else "/power1_input") for d in hw]
caps = [rd(d + "/power1_cap") for d in hw]
args = sys.argv[1:] or ["--steps", "150", "--warmup", "5", "--no-cpu-baseline", "--no-kernel-breakdown"]
child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
samples = []                                        # (t, [power of every GPU of the node], [sclk of every GPU])
t0 = time.time()
while child.poll() is None:
    samples.append((time.time() - t0, [rd(p) for p in pfiles], [rd(d + "/freq1_input") for d in hw]))
    time.sleep(0.02)
out = child.stdout.read()
import json
line = json.loads(out.strip().splitlines()[-1])
dur = line["steps"] * line["ms_per_step"] / 1e3
# which GPU is ours: the one whose power moves the most over the run (the box shows every GPU of the node; others belong to other jobs)
spread = []
for g in range(len(hw)):
    v = [s[1][g] for s in samples if s[1][g] is not None]
    spread.append((max(v) - min(v)) if v else 0)
g = max(range(len(hw)), key=lambda i: spread[i])
pw = [(t, p[g] / 1e6) for t, p, _ in samples if p[g] is not None]
best = None
for i in range(len(pw)):
    j = i
    while j < len(pw) and pw[j][0] - pw[i][0] < dur * 0.8:
        j += 1
    if j >= len(pw):
        break
    avg = sum(p for _, p in pw[i:j]) / (j - i)
    if best is None or avg > best[0]:
        best = (avg, pw[i][0], pw[j - 1][0], max(p for _, p in pw[i:j]))
fr = [f[g] / 1e6 for t, _, f in samples if f[g] is not None and best and best[1] <= t <= best[2]]
print(f"bench: {line['value']:.1f} ff/s, {line['ms_per_step']:.2f} ms/step over {line['steps']} steps ({dur:.1f} s timed)")
print(f"GPU under test = {hw[g]} (power spread over the run by GPU, W: {[round(x / 1e6) for x in spread]})")
print(f"power cap {caps[g] / 1e6 if caps[g] else None} W; busiest {0.8 * dur:.1f}-s window: mean {best[0]:.0f} W, max sample {best[3]:.0f} W; "
      f"sclk reads in it: mean {sum(fr) / max(len(fr), 1):.0f} MHz (min {min(fr) if fr else None}, max {max(fr) if fr else None}); "
      f"first sample of the run: {pw[0][1]:.0f} W; {len(pw)} samples")
