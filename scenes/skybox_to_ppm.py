#!/usr/bin/env python3
"""Convert a skybox folder of JPEGs (right/left/top/bottom/front/back.jpg, as the reference's `env`
line names it, scene.cpp:333) into binary PPMs for `p3d_render --skybox DIR`.

    python scenes/skybox_to_ppm.py /path/to/skybox  out_dir
"""
import os
import sys

from PIL import Image

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
for name in ("right", "left", "top", "bottom", "front", "back"):
    Image.open(os.path.join(src, name + ".jpg")).convert("RGB").save(os.path.join(dst, name + ".ppm"))
    print("wrote", os.path.join(dst, name + ".ppm"))
