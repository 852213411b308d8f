#!/usr/bin/env python3
"""Launch-by-launch timeline of ONE literal frame on an idle chip, from a rocprofv3 --kernel-trace of
`bench.py --frames-in-flight 1` (every frame waits for the one before).

usage: frame_timeline.py <..._kernel_trace.csv> [first_kernel_substring=clear_kernel]

A frame = the dispatches from one `clear_kernel` to the next.  Prints, averaged over the frames that have the commonest
launch sequence: each launch's start relative to the frame's first start, its duration, and the gap in front of it
(end of the previous launch -> its start).  What the critical path of a lone frame is made of."""
import csv
import re
import sys
from collections import Counter

path = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "clear_kernel"
rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("void p3d::", "").replace("p3d::", "")
    return re.sub(r"\(.*$", "", n).strip()


frames, cur = [], None
for r in rows:
    name = short(r["Kernel_Name"])
    if first in name:
        if cur:
            frames.append(cur)
        cur = []
    if cur is not None:
        cur.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
if cur:
    frames.append(cur)
sig = Counter(tuple(k[0] for k in f) for f in frames)
common, n_common = sig.most_common(1)[0]
sel = [f for f in frames if tuple(k[0] for k in f) == common]
print("%d frames in the trace, %d with the commonest sequence of %d launches" % (len(frames), n_common, len(common)))
print("%-78s %8s %9s %9s %8s" % ("launch", "groups", "start us", "dur us", "gap us"))
tot = 0.0
for i, name in enumerate(common):
    st = sum(f[i][1] - f[0][1] for f in sel) / len(sel) / 1e3
    du = sum(f[i][2] - f[i][1] for f in sel) / len(sel) / 1e3
    gap = sum((f[i][1] - f[i - 1][2]) if i else 0 for f in sel) / len(sel) / 1e3
    print("%-78s %8d %9.1f %9.1f %8.1f" % (name[:78], sel[0][i][3], st, du, gap))
end = sum(f[-1][2] - f[0][1] for f in sel) / len(sel) / 1e3
print("frame, first start -> last end: %.1f us" % end)
