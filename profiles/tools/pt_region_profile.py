import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import p3d_amd as p3d
scene = sys.argv[1]; res = int(sys.argv[2]); spp = int(sys.argv[3])
hs = p3d.HostScene(scene); hs.set_resolution(res, res)
dev = p3d.DeviceScene(hs, bvh=True)
cfg = p3d.pathtrace_config(accel=p3d.ACCEL_BVH, spp_sqrt=spp, collect_stats=0)
prof = torch.zeros(30, dtype=torch.int64, device='cuda')
assert dev._L.p3d_debug_set_pt_prof(C.c_void_p(prof.data_ptr())) == 0
rgb = torch.empty(res*res*3, dtype=torch.float32, device='cuda')
tile = p3d.Tile(0, 0, res, res, 0, 1)
dev.render_device(cfg, tile, rgb.data_ptr()); torch.cuda.synchronize()
prof.zero_(); torch.cuda.synchronize()
dev.render_device(cfg, tile, rgb.data_ptr()); torch.cuda.synchronize()
p = prof.cpu().numpy().reshape(10, 3).astype(np.float64)
names = ['loop head', 'refill/resume', 'closest #1', 'hit/miss+material+roulette', 'diffuse setup+light sample', 'closest #2 (light)', 'diffuse finish', 'mirror', 'dielectric', 'epilogue']
tot = p[:, 0].sum()
print('%-32s %8s %10s %8s' % ('region', 'time %', 'entries', 'lanes/64'))
for i, n in enumerate(names):
    if p[i, 2] > 0:
        print('%-32s %8.1f %10d %8.2f' % (n, 100 * p[i, 0] / tot, p[i, 2], p[i, 1] / p[i, 2] / 64))
