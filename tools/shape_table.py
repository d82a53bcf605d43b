#!/usr/bin/env python3
"""Per-shape kernel-time table of the training step.

    MSG_CLOCK_SHAPES=1 python bench.py --no-cpu-baseline > gpurun_out/bench_shapes.json
    python tools/shape_table.py gpurun_out/bench_shapes.json

Reads bench.py's JSON line (its "kernels" object, keyed per kernel and -- with MSG_CLOCK_SHAPES=1 -- per problem
shape) and prints the entries sorted by total time per training iteration.
"""
import json
import sys

line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
steps = d.get("clock_iterations") or d["steps"]         # (bench.py --clock-only plain|regularised: fewer than steps)
rows = []
for k, v in d["kernels"].items():
    rate = v.get("TFLOP/s", v.get("GB/s"))
    unit = "TFLOP/s" if "TFLOP/s" in v else "GB/s"
    rows.append((v["launches"] * v["avg_us"] / steps / 1e3, v["launches"] / steps, v["avg_us"], rate, unit, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{d['value']:.2f} {d['unit']}, {d['ms_per_step']:.1f} ms/step; timed kernels: {tot:.1f} ms/step")
print(f"{'ms/step':>8} {'calls/step':>10} {'avg us':>9} {'rate':>8}  kernel | shape")
for ms, calls, avg, rate, unit, k in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print(f"{ms:8.2f} {calls:10.1f} {avg:9.1f} {rate:8.1f} {unit:8s} {k}")
