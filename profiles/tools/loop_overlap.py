#!/usr/bin/env python3
"""What overlaps what in bench.py's timed loop: from a rocprofv3 kernel trace of `bench.py --steps N` (frames in flight as bench.py
keeps them), over the longest run of back-to-back literal frames (the timed loops), print per kernel the mean duration under load, per
hardware queue what it carried, and for how much of the time 0 / 1 / 2 / 3+ pass-1 kernels were in flight.

CAVEAT: with several streams in flight the tracer itself changes the picture (a 0.10-ms frame of the loop takes 0.17-0.30 ms under
`rocprofv3 --kernel-trace`, and most of the time no pass-1 kernel is in flight): the durations and the overlap histogram of such a trace
say little about the untraced loop.  What it does show reliably is WHICH hardware queue carried which launches.

usage: python3 profiles/tools/loop_overlap.py <..._kernel_trace.csv> [frames to look at, default 400]"""
import csv
import re
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
want = int(sys.argv[2]) if len(sys.argv) > 2 else 400


def short(name):
    m = re.match(r"(?:void )?(?:p3d::)?(\w+)(?:<(.*)>)?", name)
    base, args = m.group(1), (m.group(2) or "")
    if base == "whitted_kernel":
        a = [x.strip() for x in args.split(",")]
        return "whitted LIT=%s%s" % (a[6], " counting" if a[2] == "true" else "")
    return base


def is_pass1(name):
    return short(name) == "whitted LIT=1"


# the window: the last `want` pass-1 launches of the densest stretch (the last timed loop)
p1 = [i for i, r in enumerate(rows) if is_pass1(r[2])]
if len(p1) < 8:
    sys.exit("no literal frames in this trace")
lo = rows[p1[max(0, len(p1) - want - 8)]][0]
hi = rows[p1[-8]][0]  # (leave the drain of the last frames out)
win = [r for r in rows if r[0] >= lo and r[1] <= hi]
span = (hi - lo) / 1e3
frames = sum(1 for r in win if is_pass1(r[2]))
print("window: %.1f us, %d frames -> %.4f ms per frame" % (span, frames, span / frames / 1e3))
dur = defaultdict(list)
per_q = defaultdict(lambda: defaultdict(int))
for s, e, n, q in win:
    dur[short(n)].append((e - s) / 1e3)
    per_q[q][short(n)] += 1
print("%-28s %8s %10s %12s" % ("kernel", "launches", "mean us", "busy / span"))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-28s %8d %10.1f %12.2f" % (k, len(v), sum(v) / len(v), sum(v) / span))
print("hardware queues:")
for q in sorted(per_q):
    print("  queue %s: %s" % (q, ", ".join("%s x%d" % kv for kv in sorted(per_q[q].items()))))
ev = []
for s, e, n, q in win:
    if is_pass1(n):
        ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist = defaultdict(float)
cur, last = 0, lo
for t, d in ev:
    hist[min(cur, 3)] += t - last
    cur += d; last = t
hist[min(cur, 3)] += hi - last
tot = sum(hist.values())
print("pass-1 kernels in flight: " + ", ".join("%s: %.2f" % (("3+" if k == 3 else k), v / tot) for k, v in sorted(hist.items())))
