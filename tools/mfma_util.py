#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 PMC pass (north star: "rocprof reports MFMA utilisation for the modulated convs
against gfx950 peak"):

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
              SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d <dir> -- python <program>
    python tools/mfma_util.py <dir> [pattern ...]

SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over all SIMDs (MI355X_MICROARCH.md: 32 x the MFMA count for
32x32x16 bf16, 16 x for 16x16x32: checked here against SQ_INSTS_MFMA); GRBM_GUI_ACTIVE is the launch's busy time in shader
clock cycles SUMMED OVER THE 8 XCDs (calibrated on launches of known duration: 15.5 "GHz" = 8 x 1.93).  Utilisation =
busy / (GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): the share of all matrix-pipe cycles of the chip AT THE CLOCK IT HELD during the
launch in which an MFMA was executing; times (held clock / 2.4 GHz) it is the fraction of the quoted dense peak.  The SQ_*
wave counters are quad-cycles; they are printed as shares of SQ_WAVE_CYCLES."""
import collections
import csv
import glob
import sys

root, pats = sys.argv[1], sys.argv[2:] or ["conv_", "nl_attn"]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row.get("Kernel_Name", "")
        if not any(p in name for p in pats):
            continue
        key = name.split("(")[0][:70]
        acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(key, row["Counter_Name"])] += 1
print(f"{'kernel':72s} {'launches':>8s} {'MFMA util':>10s} {'MFMA insts':>12s} {'wait_any':>9s} {'wait_inst':>9s} {'active':>8s}")
for key, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
    n = cnt[(key, "GRBM_GUI_ACTIVE")] or 1
    busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0:
        continue
    wave = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    print(f"{key:72s} {n:8d} {busy / (gui / 8 * 256 * 4):10.3f} {c.get('SQ_INSTS_MFMA', 0.0) / n:12.3e} "
          f"{c.get('SQ_WAIT_ANY', 0.0) / wave:9.3f} {c.get('SQ_WAIT_INST_ANY', 0.0) / wave:9.3f} {c.get('SQ_ACTIVE_INST_ANY', 0.0) / wave:8.3f}")
