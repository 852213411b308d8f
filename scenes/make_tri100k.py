#!/usr/bin/env python3
"""Generate the synthetic triangle-soup scene of BASELINE.json configs[3] (SURVEY.md §8(d) cfg 4).

100 000 random triangles: centres U[-1,1]^3 (float32), each vertex = centre +
U[-0.03,0.03]^3, printed with %.6f; numpy default_rng(20261003).  The text is
~8.9 MB, so it is generated on demand (tests, bench) instead of being committed.

    python scenes/make_tri100k.py OUT.p3f [n_triangles] [resolution]
"""
import sys

import numpy as np

HEADER = """bclr 0.078 0.361 0.753
v
from 0 0 4.5
at 0 0 0
up 0 1 0
angle 35
hither 0.01
resolution {res} {res}
aperture 0
focal 1
l 4 3 5 1 1 1
l -3 4 4 1 1 1
f 1 0.9 0.7 0.5 1 1 1 0.5 30.0827 0 1 0 0 0
"""


def generate(path, n=100000, res=512, seed=20261003):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)
    offs = rng.uniform(-0.03, 0.03, (n, 3, 3)).astype(np.float32)
    verts = centres[:, None, :] + offs
    with open(path, "w") as f:
        f.write(HEADER.format(res=res))
        lines = []
        for t in range(n):
            lines.append("p 3\n")
            for k in range(3):
                lines.append("%.6f %.6f %.6f\n" % tuple(float(v) for v in verts[t, k]))
        f.write("".join(lines))
    return path


if __name__ == "__main__":
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    res = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    generate(out, n, res)
