#!/usr/bin/env python3
"""Shader clock and power of GPU 0 while a command runs (rocm-smi polled beside it):  python tools/clock_watch.py -- python bench.py ...
Prints the distribution of the sampled sclk / power over the run's busy phase (samples with >= 90 % GPU use)."""
import json, subprocess, sys, time

cmd = sys.argv[sys.argv.index("--") + 1:]
proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
samples = []
while proc.poll() is None:
    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showuse", "--json"], capture_output=True,
                             text=True, timeout=5).stdout
        card = next(iter(json.loads(out).values()))
        sclk = next((v for k, v in card.items() if k.startswith("sclk")), "")
        mhz = float(sclk.strip("()").lower().replace("mhz", "")) if sclk else float("nan")
        power = next((float(v) for k, v in card.items() if "Power" in k and "W" in k), float("nan"))
        use = next((float(v) for k, v in card.items() if k.startswith("GPU use")), float("nan"))
        samples.append((mhz, power, use))
    except Exception as e:                                   # (the tool is best effort: a missing field must not kill the run)
        samples.append((float("nan"), float("nan"), float("nan")))
    time.sleep(0.1)
print(proc.stdout.read()[-400:])
busy = [s for s in samples if s[2] == s[2] and s[2] >= 90] or samples
for name, idx in (("sclk MHz", 0), ("power W", 1)):
    vals = sorted(v[idx] for v in busy if v[idx] == v[idx])
    if vals:
        print(f"{name}: n={len(vals)} min {vals[0]:.0f}  median {vals[len(vals) // 2]:.0f}  mean {sum(vals) / len(vals):.0f}  max {vals[-1]:.0f}")
print(f"samples {len(samples)}, busy {len(busy)}")
