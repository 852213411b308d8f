#!/usr/bin/env python3
"""Condense the raw output of profiles/r01_profile_recipe.sh (under gpurun_out/) into the
summaries kept under profiles/<round>/.  usage: summarize_profile.py gpurun_out/prof_r01c profiles/r01"""
import csv, glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)

for sub, name in (("trace", "cfg2"), ("trace_tri", "tri100k"), ("trace_pt", "cornell_pt")):
    (stats,) = glob.glob(f"{src}/{sub}/*/*_kernel_stats.csv")
    shutil.copy(stats, f"{dst}/{name}_kernel_stats.csv")
for f, name in (("bench_trace.json", "cfg2_bench_under_rocprof.json"), ("bench_tri.json", "tri100k_bench_under_rocprof.json"),
                ("bench_pt.json", "cornell_pt_bench_under_rocprof.json"), ("bench_plain.json", "cfg2_bench_plain.json")):
    shutil.copy(f"{src}/{f}", f"{dst}/{name}")

KERNEL = "whitted_kernel<2, true, false, false"  # the timed kernel of the default bench (any SPILL argument)
vals = {}
for path in glob.glob(f"{src}/pmc_*/*/*_counter_collection.csv"):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if KERNEL not in row["Kernel_Name"]:
                continue
            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
summary = {k: {"max": max(v), "mean": sum(v) / len(v), "min": min(v), "n": len(v)} for k, v in sorted(vals.items())}
with open(f"{dst}/cfg2_pmc_summary.json", "w") as fh:
    json.dump(summary, fh, indent=1)

fetch_kb, write_kb = summary["FETCH_SIZE"]["mean"], summary["WRITE_SIZE"]["mean"]
hbm = {
    "kernel": "whitted_kernel<2,true,false,false>",
    "workload": "cfg2 (balls_low 1024x1024 depth 4 BVH), outputs rgb f32 + hit i32 + rgb8",
    "FETCH_SIZE_KB_mean": fetch_kb,
    "WRITE_SIZE_KB_mean": write_kb,
    "correction": "FETCH_SIZE doubled (gfx950 reports half of a coalesced stream, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is",
    "hbm_bytes_per_launch": int(round((2 * fetch_kb + write_kb) * 1024)),
    "source": f"{dst}/cfg2_pmc_summary.json (separate --pmc passes, profiles/r01_profile_recipe.sh)",
}
with open(os.path.join(os.path.dirname(dst.rstrip('/')) or ".", os.path.basename(dst.rstrip('/')) + "_pmc_hbm_bytes.json"), "w") as fh:
    json.dump(hbm, fh, indent=1)
print(json.dumps(hbm))
