"""Random .p3f scenes for differential testing (GPU kernels vs CPU oracle)."""
import numpy as np


def random_scene(seed, path, n_spheres=6, n_tris=8, n_boxes=2, n_planes=0, n_lights=2, res=(96, 80), emitters=0):
    rng = np.random.default_rng(seed)

    def v(lo, hi, n=3):
        return " ".join("%.6g" % x for x in rng.uniform(lo, hi, n))

    def material(kind):
        cd, cs = v(0.05, 1), v(0, 1)
        if kind == "glass":
            return "f %s %.3g %s %.3g %.4g 1 %.3g 0 0 0" % (cd, rng.uniform(0, 0.6), cs, rng.uniform(0, 0.9),
                                                            rng.uniform(5, 80), rng.uniform(1.05, 1.8))
        if kind == "matte":
            return "f %s %.3g %s 0 %.4g 0 1 0 0 0" % (cd, rng.uniform(0.3, 1), cs, rng.uniform(5, 80))
        if kind == "pt_diffuse":
            return "f %s 1 0 0 0 0 10 0 1 0 0 0" % cd
        if kind == "pt_mirror":
            return "f %s 0 %s 1 300 0 1 0 0 0" % (cd, cs)
        if kind == "pt_glass":
            return "f %s 0 %s 0 300 1 %.3g 0 0 0" % (cd, cs, rng.uniform(1.2, 1.7))
        if kind == "emitter":
            return "f 0 0 0 1 0 0 0 0 10 0 1 %s" % v(3, 12)
        return "f %s %.3g %s %.3g %.4g 0 1 0 0 0" % (cd, rng.uniform(0.3, 1), cs, rng.uniform(0.05, 0.9), rng.uniform(5, 80))

    pt = emitters > 0
    kinds = ["pt_diffuse", "pt_diffuse", "pt_mirror", "pt_glass"] if pt else ["shiny", "shiny", "matte", "glass"]
    lines = ["bclr %s" % v(0, 0.8), "v", "from %s" % v(3, 5), "at %s" % v(-0.3, 0.3),
             "up 0 %s" % ("1 0" if rng.random() < 0.5 else "0 1"), "angle %.4g" % rng.uniform(30, 60), "hither 0.01",
             "resolution %d %d" % res, "aperture %.3g" % (rng.uniform(2, 12) if rng.random() < 0.3 else 0),
             "focal %.3g" % rng.uniform(0.8, 1.2)]
    for _ in range(n_lights):
        lines.append("l %s %s" % (v(-6, 6), v(0.3, 1)))
    objs = (["s"] * n_spheres) + (["p"] * n_tris) + (["box"] * n_boxes) + (["pl"] * n_planes)
    rng.shuffle(objs)
    for o in objs:
        if rng.random() < 0.6 or not any(ln.startswith("f ") for ln in lines):
            lines.append(material(rng.choice(kinds)))
        if o == "s":
            lines.append("s %s %.4g" % (v(-1.5, 1.5), rng.uniform(0.15, 0.6)))
        elif o == "p":
            c = rng.uniform(-1.5, 1.5, 3)
            pts = [c + rng.uniform(-0.9, 0.9, 3) for _ in range(3)]
            lines.append("p 3\n" + "\n".join(" ".join("%.6g" % x for x in p) for p in pts))
        elif o == "box":
            lo = rng.uniform(-1.8, 1.2, 3)
            lines.append("box %s %s" % (" ".join("%.6g" % x for x in lo), " ".join("%.6g" % x for x in lo + rng.uniform(0.2, 0.8, 3))))
        else:
            pts = [rng.uniform(-2, 2, 3) for _ in range(3)]
            lines.append("pl " + "  ".join(" ".join("%.6g" % x for x in p) for p in pts))
    for _ in range(emitters):
        lines.append(material("emitter"))
        lines.append("s %s %.4g" % (v(-1.5, 1.5), rng.uniform(0.1, 0.3)))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path
