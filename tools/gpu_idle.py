#!/usr/bin/env python3
"""How much of a training iteration the GPU sits idle: union of the kernel intervals of a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/idle -- python bench.py --no-cpu-baseline --no-kernel-clock
    python tools/gpu_idle.py gpurun_out/idle [last_ms]

Prints, for the last `last_ms` milliseconds of the trace (default 1000: timed iterations only): wall time, time covered by at least one kernel, the idle
remainder, and the largest gaps with the kernels on either side of them (run on the GPU box: the traces are large).
"""
import csv
import glob
import os
import sys

path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
cut = t1 - int(float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 1e9)
rows = [r for r in rows if r[0] >= cut]
busy, cur_end, gaps = 0, rows[0][0], []
prev_name = ""
for s, e, n in rows:
    if s > cur_end:
        gaps.append((s - cur_end, prev_name, n))
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
    if e >= cur_end:
        prev_name = n
wall = rows[-1][1] - rows[0][0]
print(f"kernels {len(rows)}  wall {wall / 1e6:.1f} ms  busy {busy / 1e6:.1f} ms  idle {(wall - busy) / 1e6:.1f} ms ({100 * (wall - busy) / wall:.1f} %)")
small = sum(g for g, _, _ in gaps if g < 20000)
print(f"gaps < 20 us: {sum(1 for g, _, _ in gaps if g < 20000)} totalling {small / 1e6:.1f} ms; larger: "
      f"{sum(1 for g, _, _ in gaps if g >= 20000)} totalling {(wall - busy - small) / 1e6:.1f} ms")
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print(f"  {g / 1e3:8.1f} us   after {a[:70]:70s} before {b[:70]}")
