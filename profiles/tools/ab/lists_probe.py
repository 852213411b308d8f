import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["P3D_PRINT_HANDOFF"] = "1"
import p3d_amd as p3d
hs = p3d.HostScene("tests/golden/scenes/balls_low.p3f"); hs.set_resolution(1024, 1024)
dev = p3d.DeviceScene(hs, bvh=True)
for i in range(2):
    rgb, hit, st = dev.render(p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=4, collect_stats=1))
    print(st.handoff_checked, st.handoff_redone, st.handoff_rounds, st.kernel_ms, st.pass1_ms, st.handoff_ms)
