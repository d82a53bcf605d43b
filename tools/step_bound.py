#!/usr/bin/env python3
"""How far each launch of the training step is from max(compute, memory) -- a map of where the remaining time is.

    MSG_CLOCK_SHAPES=1 python bench.py --no-cpu-baseline > gpurun_out/bench_shapes.json      (GPU box)
    python tools/step_bound.py gpurun_out/bench_shapes.json [--mfma 1400] [--hbm 5000]

For every timed launch the floor is max(algorithmic FLOPs / R_mfma, algorithmic bytes / R_hbm) with R_mfma = what the dominant
kernel SUSTAINS in this step (default 1 400 TFLOP/s: the clock the chip holds under bf16 MFMA streams) and
R_hbm = what the streaming kernels sustain (default 5 000 GB/s: 80 % of the measured copy rate) -- NOT the quoted peaks: the
question here is which launches are far from what this chip has been seen to do, not from a datasheet.  Contraction launches:
FLOPs from the timing key's work, bytes = input + output maps (+ per-sample weight sets) parsed from the shape key.  Streaming
launches: bytes from the key's work.  Launches without a shape key (few-row linears, tiny fp32 kernels) count at their
measured time.  Prints the step's total, its floor, and the families / shapes that own the gap."""
import collections
import json
import re
import sys

path = sys.argv[1]
opt = lambda name, dflt: float(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else dflt
R_MFMA, R_HBM = opt("--mfma", 1400.0) * 1e12, opt("--hbm", 5000.0) * 1e9
d = json.loads([l for l in open(path) if l.startswith("{")][-1])
steps = d.get("clock_iterations") or d["steps"]
SHAPE = re.compile(r"B(\d+) (\d+)x(\d+)->(\d+)x(\d+) (\d+)->(\d+) (\d)x(\d) s(\d)( up\d)?( ps)?( per-sample| shared)?")
rows = []
for key, v in d["kernels"].items():
    t_ms = v["launches"] * v["avg_us"] / steps / 1e3
    fam = key.split("|")[0].split("/")[0]
    floor_us = v["avg_us"]
    if "TFLOP/s" in v and "|" in key:
        m = SHAPE.search(key)
        flops = v["TFLOP/s"] * 1e12 * v["avg_us"] * 1e-6
        b, ih, iw, oh, ow, ci, co, kh, kw, st = (int(m.group(i)) for i in range(1, 11))
        ps, per_sample = bool(m.group(12)), (m.group(13) or "").strip() == "per-sample"
        esz = 2
        if fam == "conv_wgrad":                                   # reads gy (co channels at the output size) and x
            gy_hw = (2 * ih * 2 * iw) if ps else oh * ow
            byts = b * (gy_hw * co + ih * iw * ci) * esz + (b if per_sample else 1) * co * ci * kh * kw * 4
        else:
            out_hw = (2 * oh * 2 * ow) if ps else oh * ow
            out_c = co // 4 if ps else co
            byts = b * (ih * iw * ci + out_hw * out_c) * esz + (b if per_sample else 1) * co * ci * kh * kw * esz
        floor_us = max(flops / R_MFMA, byts / R_HBM) * 1e6
    elif "GB/s" in v and v["GB/s"] > 200:                        # a streaming launch worth a floor (tiny fp32 ones: measured time)
        floor_us = v["GB/s"] * 1e9 * v["avg_us"] * 1e-6 / R_HBM * 1e6
    floor_us = min(floor_us, v["avg_us"])
    rows.append((t_ms, floor_us * v["launches"] / steps / 1e3, fam, key))
tot, flo = sum(r[0] for r in rows), sum(r[1] for r in rows)
print(f"{d['value']} {d['unit']}; timed launches {tot:.1f} ms per iteration, floor at {R_MFMA / 1e12:.0f} TFLOP/s / {R_HBM / 1e9:.0f} GB/s: "
      f"{flo:.1f} ms ({100 * flo / tot:.0f} %), gap {tot - flo:.1f} ms")
fam = collections.defaultdict(lambda: [0.0, 0.0])
for t, f, fa, _ in rows:
    fam[fa][0] += t; fam[fa][1] += f
print(f"\n{'family':28s} {'ms':>7s} {'floor':>7s} {'gap':>6s}")
for fa, (t, f) in sorted(fam.items(), key=lambda kv: -(kv[1][0] - kv[1][1])):
    if t - f > 0.05:
        print(f"{fa:28s} {t:7.2f} {f:7.2f} {t - f:6.2f}")
print(f"\n{'ms':>6s} {'floor':>6s} {'gap':>5s}  launch")
for t, f, _, key in sorted(rows, key=lambda r: -(r[0] - r[1]))[:28]:
    print(f"{t:6.2f} {f:6.2f} {t - f:5.2f}  {key}")
