#!/usr/bin/env python3
"""Condense the raw output of profiles/r04_profile_recipe.sh (gpurun_out/prof_r04/<workload>_<mode>/) into
profiles/r04/<workload>_<mode>_{kernel_stats.csv, pmc_summary.json, bench_*.json}.

usage: summarize_r04.py <workload> <literal|per_pixel>

The summary carries the SHA-256 of the device sources it was collected from (bench.kernel_source_hash): bench.py quotes
its counters only while that hash matches the tree.  Per kernel: calls, average duration (kernel-trace --stats) and the
mean of every counter per dispatch.  `dominant` = the kernel the roofline block of bench.py is about: the timed
instantiation of whitted_kernel / pt_kernel with the largest total duration.  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE
KiB (gfx950 reports half of a coalesced read stream: MI355X_MICROARCH.md, HBM)."""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

workload, mode = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "prof_r04", "%s_%s" % (workload, mode))
dst = os.path.join(ROOT, "profiles", "r04")
os.makedirs(dst, exist_ok=True)
tag = "%s_%s" % (workload, mode)


def short(name):
    return re.sub(r"\(.*$", "", name.replace("void p3d::", "").replace("p3d::", "")).strip()


def newest(pattern):
    """a pass directory may hold the output of an earlier run of the recipe as well (gpurun merges): take the latest"""
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime) if files else None


stats_csv = newest(src + "/trace/*/*_kernel_stats.csv")
shutil.copy(stats_csv, os.path.join(dst, tag + "_kernel_stats.csv"))
for f in ("bench_under_rocprof.json", "bench_plain.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, tag + "_" + f))
kernels = {}
for row in csv.DictReader(open(stats_csv)):
    kernels[short(row["Name"])] = {"calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6,
                                    "total_ms": float(row["TotalDurationNs"]) / 1e6, "counters": {}}
vals = {}
for pass_dir in sorted(glob.glob(src + "/pmc_*")):
    path = newest(pass_dir + "/*/*_counter_collection.csv")
    if not path:
        continue
    for row in csv.DictReader(open(path)):
        vals.setdefault((short(row["Kernel_Name"]), row["Counter_Name"]), []).append(float(row["Counter_Value"]))
for (k, c), v in sorted(vals.items()):
    kernels.setdefault(k, {"counters": {}})["counters"][c] = {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "n": len(v)}

# the timed instantiation: no counters (3rd template argument false); literal launches = ..., 1, 1>, per-pixel = ..., 0>
want = r"(whitted_kernel<\d, (true|false), false, .*, 1, (true|false)>$)" if mode == "literal" else r"(whitted_kernel<\d, (true|false), false, .*, 0, (true|false)>$)"
cands = [k for k in kernels if (re.search(want, k) or k.startswith("pt_kernel<")) and "avg_ms" in kernels[k] and ", true, true" not in k[:40]]
cands = [k for k in cands if not re.match(r"(whitted|pt)_kernel<\d, (true|false), true", k)]
dom = max(cands, key=lambda k: kernels[k]["total_ms"])
c = kernels[dom]["counters"]
g = lambda n: c[n]["mean"] if n in c else None
dominant = {"kernel": dom, "avg_ms": round(kernels[dom]["avg_ms"], 5), "calls": kernels[dom]["calls"]}
for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
          "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_LDS_BANK_CONFLICT",
          "FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "GRBM_GUI_ACTIVE"):
    if g(n) is not None:
        dominant[n] = g(n)
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    dominant["hbm_bytes"] = int(round((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024))
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_INSTS_VALU"):
    # active lanes per VALU wave-instruction / 64 (the round-1 definition, VERDICT.md quotes it)
    dominant["lane_utilisation"] = round(g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_INSTS_VALU")), 4)
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
    dominant["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
out = {"workload": workload, "stack_mode": mode, "source_hash": bench.kernel_source_hash(),
       "recipe": "profiles/r04_profile_recipe.sh %s %s" % (workload, mode), "dominant": dominant, "kernels": kernels}
json.dump(out, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(dominant, indent=1))
