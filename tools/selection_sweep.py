#!/usr/bin/env python3
"""Reads the runs of tools/selection_sweep.sh (gpurun_out/sweep_*.json): for every (direction, shape) of the training step, the time
per iteration under each forced kernel selection against the better of two default runs; lists what would be > 5 % faster."""
import json

NAMES = ["variant1", "variant2", "pp0", "shortk0", "narrow0", "upconv0", "w32_0", "wgrow3_0", "wgw32_0", "lean0"]


def load(name):
    d = json.loads(open(f"gpurun_out/sweep_{name}.json").read().strip().splitlines()[-1])
    steps = d.get("clock_iterations") or d["steps"]
    out = {}
    for k, v in d["kernels"].items():
        if "|" not in k:
            continue
        fam, shape, _ = k.split("|")
        kind = "wgrad" if fam.startswith("conv_wgrad") else ("actbwd" if "actbwd" in fam else "fprop")
        e = out.setdefault((kind, shape), [0.0, fam])
        e[0] += v["launches"] * v["avg_us"] / steps / 1e3
    return d, out


d0, base = load("s0")
d1, base2 = load("s0b")
print(f"default: {d0['value']} / {d1['value']} img/s, plain {d0['timed_region']['plain_ms']} / {d1['timed_region']['plain_ms']} ms "
      f"({d0['library']['file']})")
for n in NAMES:
    d, o = load(n)
    wins, loss = [], 0.0
    for key, (t, fam) in o.items():
        if key in base and key in base2:
            lo, hi = min(base[key][0], base2[key][0]), max(base[key][0], base2[key][0])
            if t < 0.95 * lo and lo - t > 0.01:
                wins.append((lo - t, key, base[key][1], fam, lo, t))
            if t > 1.05 * hi:
                loss += t - hi
    wins.sort(reverse=True)
    print(f"== {n}: {d['value']} img/s, plain {d['timed_region']['plain_ms']} ms; shapes > 5 % faster: {len(wins)} "
          f"(-{sum(w[0] for w in wins):.2f} ms per iteration), slower shapes: +{loss:.2f} ms")
    for w in wins[:10]:
        print(f"     -{w[0]:.3f} ms  {w[1][0]} {w[1][1]}:  {w[2]} {w[4]:.3f} -> {w[3]} {w[5]:.3f} ms")
