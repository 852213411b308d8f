import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scenes"))
import p3d_amd as p3d, make_tri100k
p = "/tmp/tri100k_probe.p3f"
make_tri100k.generate(p)
hs = p3d.HostScene(p); hs.set_resolution(2048, 2048)
dev = p3d.DeviceScene(hs, bvh=True)
for d in (1, 6):
    cfg = p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=d, collect_stats=1, stack_mode=p3d.STACK_PER_PIXEL)
    rgb, hit, st = dev.render(cfg)
    steps, uni, lanes, leaf, toplanes = st.plane_tests, st.box_tests, st.sphere_tests, st.rays_bounce, st.rays_light
    print("depth", d, "rays", st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract, "node_tests", st.node_tests, "tri_tests", st.tri_tests)
    print("  wave descend steps", steps, "uniform", uni, "= %.3f" % (uni / max(steps, 1)), "active lanes per step %.1f" % (lanes / max(steps, 1)),
          "lane-visits in the first 63 slots %.3f" % (toplanes / max(lanes, 1)), "wave leaf fetches", leaf)
