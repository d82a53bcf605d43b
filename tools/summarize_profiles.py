#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs of tools/collect_profiles.sh into the committed summaries:
   profiles/<tag>_kernel_stats.md (+ .csv)  and  profiles/<tag>_pmc_traffic.json.
usage: python tools/summarize_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"profiles_{tag}")
dst = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--dst" else os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def bench_line(path):
    for line in open(path):
        if line.startswith("{\"metric\""):
            return json.loads(line)
    return None


def short(name):
    name = name.replace("at::native::", "").replace("(anonymous namespace)::", "").replace("unsigned short", "bf16")
    return re.sub(r"\(.*", "", name)[:110]


KEYS = [("conv_fprop_row3_kernel<4, 4, true, 1", "conv_fprop_row3_actbwd/bf16"), ("conv_fprop_row3_kernel<2, 2, true, 1", "conv_fprop_row3n_actbwd/bf16"),
        ("conv_fprop_row3_kernel<4, 4", "conv_fprop_row3/bf16"), ("conv_fprop_row3_kernel<2, 2", "conv_fprop_row3n/bf16"), ("conv_fprop_pp_kernel", "conv_fprop_pp/bf16"), ("conv_fprop_kernel<unsigned short, true", "conv_fprop_dma/bf16"),
        ("conv_fprop_kernel<unsigned short, false", "conv_fprop_reg/bf16"),
        ("conv_wgrad_row3_kernel", "conv_wgrad_row3/bf16"), ("conv_wgrad_row3s_kernel", "conv_wgrad_row3s/bf16"),
        ("conv_upconv_kernel", "conv_fprop_upconv/bf16"),
        ("conv_wgrad_kernel<unsigned short", "conv_wgrad/bf16"),
        ("nl_attn_fwd_kernel<unsigned short", "nl_attention_fwd/bf16"),
        ("nl_attn_bwd_q_kernel<unsigned short", "nl_attention_bwd_q/bf16"),
        ("nl_attn_bwd_kv_kernel<unsigned short", "nl_attention_bwd_kv/bf16"),
        ("mbstd_fwd_kernel<unsigned short", "mbstd_fwd/bf16"),
        ("mbstd_bwd_kernel<unsigned short", "mbstd_bwd/bf16"),
        ("upfirdn2d_vec_kernel<unsigned short, 1, 1", "upfirdn2d/bf16/up1down1/vec"),
        ("blur_sep_kernel<unsigned short, 32, 3, true>", "upfirdn2d/bf16/up1down1/sep+act"),
        ("blur_sep_kernel<unsigned short, 16, 3, true>", "upfirdn2d/bf16/up1down1/sep+act"),
        ("blur_sep_kernel<unsigned short, 32, 4, true>", "upfirdn2d/bf16/up1down1/sep+act"),
        ("blur_sep_kernel<unsigned short, 16, 4, true>", "upfirdn2d/bf16/up1down1/sep+act"),
        ("rgb_skip_fwd_kernel", "rgb_skip_merge/f32"), ("rgb_skip_bwd_kernel", "rgb_skip_merge_bwd/f32"),
        ("blur_sep_kernel<unsigned short", "upfirdn2d/bf16/up1down1/sep"),
        ("bias_act_vec_kernel<unsigned short", "bias_act_fwd/torch.bfloat16"),
        ("bias_act_bwd_cl_kernel<unsigned short", "bias_act_bwd/torch.bfloat16")]

stats = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(stats)))
total = sum(int(r["TotalDurationNs"]) for r in rows)
plain, prof = bench_line(os.path.join(src, "bench_plain.log")), bench_line(os.path.join(src, "bench_stats.log"))
iters = prof["steps"] + prof["warmup"]
with open(os.path.join(dst, f"{tag}_kernel_stats.md"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline` ({tag})\n\n")
    f.write(f"1x MI355X, {plain['config']['workload'].split(';')[0]}, {plain['dtype']}; {iters} iterations in the trace.  Un-profiled run of the same command "
            f"in the same gpurun call: {plain['value']} img/s ({plain['ms_per_step']} ms/step); under the profiler: "
            f"{prof['value']} img/s ({prof['ms_per_step']} ms/step).  Total kernel time {total / 1e9:.3f} s = "
            f"{total / 1e6 / iters:.1f} ms per iteration.\n\n")
    r = plain["roofline"]
    f.write(f"bench.py roofline leg (HIP events, un-profiled run): `{r['kernel']}` {r['launches']} launches in the timed "
            f"steps, avg {r['avg_us']} us, {r['achieved']} {r['unit']} = {100 * r['frac']:.1f} % of {r['peak']}.\n\n")
    f.write("| % | calls | avg us | ms / iteration | kernel |\n|---|---|---|---|---|\n")
    for r in rows[:40]:
        f.write(f"| {float(r['Percentage']):.2f} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | "
                f"{int(r['TotalDurationNs']) / 1e6 / iters:.2f} | `{short(r['Name'])}` |\n")
    stock = sum(int(r["TotalDurationNs"]) for r in rows
                if any(k in r["Name"] for k in ("at::native", "Cijk_", "rocprim", "__amd_rocclr")))
    f.write(f"\nStock torch / library kernels (`at::native::*`, `Cijk_*`, rocprim, runtime fills): {stock / 1e6 / iters:.1f} ms per "
            f"iteration = {100 * stock / total:.1f} % of kernel time (the regularised iterations in the trace run composite "
            f"torch-op formulations for their second-order graphs).\n")
import shutil
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))

traffic = {}
for leg, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    path = glob.glob(os.path.join(src, leg, "*", "*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for needle, key in KEYS:
            if needle in r["Kernel_Name"]:
                agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
                break
    for key, (s, n) in agg.items():
        traffic.setdefault(key, {})[leg] = (s, n)
out = {"_about": "HBM-side traffic per launch from rocprofv3 PMC counters, one pass per counter (FETCH_SIZE; WRITE_SIZE) with "
                 "--kernel-trace only, command `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-clock` "
                 f"(1x MI355X, {plain['config']['workload'].split(';')[0]}; averaged over every launch of the kernel in the run).  FETCH_SIZE [KiB] "
                 "is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B); WRITE_SIZE is exact.",
       "tag": tag, "kernels": {}}
for key, legs in traffic.items():
    if "fetch" in legs and "write" in legs:
        (fs, fn), (ws, wn) = legs["fetch"], legs["write"]
        out["kernels"][key] = {"launches": fn, "fetch_kib_raw_sum": fs, "write_kib_sum": ws,
                               "traffic_bytes_per_launch": round((2 * fs / fn + ws / wn) * 1024)}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
