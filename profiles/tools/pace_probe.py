#!/usr/bin/env python3
"""Which queues pace bench.py's bulk / tail loop (experiments r04 section 19): the loop with three timing events per frame.
usage (GPU box, from the repository root): python3 profiles/tools/pace_probe.py [frames in flight, default 6]"""
import os, sys, time, numpy as np, torch
torch.cuda.init()
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
import p3d_amd as p3d
hs = p3d.HostScene(os.path.join(ROOT, "tests/golden/scenes/balls_low.p3f")); hs.set_resolution(1024, 1024)
cfg = p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=4)
nfl = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bulk = [torch.cuda.current_stream(), torch.cuda.Stream()]
tails = [torch.cuda.Stream(), torch.cuda.Stream()]
scenes = [p3d.DeviceScene(hs, bvh=True) for _ in range(nfl)]
n = 1024 * 1024
bufs = [torch.empty(n * 16, dtype=torch.uint8, device="cuda") for _ in range(nfl)]
tile = scenes[0].full_tile()
for k, sc in enumerate(scenes):
    sc.render(cfg)  # tile schedule
    sc.set_tail_stream(tails[k % 2])
def frame(i, ev=None):
    k = i % nfl
    b, t = bulk[k % 2], tails[k % 2]
    if ev is not None:
        e0 = torch.cuda.Event(enable_timing=True); e0.record(b)
    scenes[k].render_device(cfg, tile, d_rgb=bufs[k].data_ptr(), d_hit=bufs[k].data_ptr() + n * 12, stream=b.cuda_stream)
    if ev is not None:
        e1 = torch.cuda.Event(enable_timing=True); e1.record(b)
        e2 = torch.cuda.Event(enable_timing=True); e2.record(t)
        ev.append((e0, e1, e2))
for i in range(60): frame(i)
torch.cuda.synchronize()
# untimed-by-events reference
t0 = time.perf_counter()
for i in range(600): frame(i)
torch.cuda.synchronize(); ref = (time.perf_counter() - t0) / 600 * 1e3
ev = []
base = torch.cuda.Event(enable_timing=True); base.record(bulk[0])
t0 = time.perf_counter()
for i in range(600): frame(i, ev)
torch.cuda.synchronize(); with_ev = (time.perf_counter() - t0) / 600 * 1e3
T = np.array([[base.elapsed_time(e) for e in trio] for trio in ev])  # ms: bulk reaches frame, pass 1 done, tail done
print("ms per frame: %.4f without events, %.4f with" % (ref, with_ev))
s = slice(100, 560)
p1 = (T[:, 1] - T[:, 0])[s]
print("bulk queue: clear + pass 1 of a frame %.4f ms mean (min %.4f max %.4f); gap to the previous frame on the queue %.4f" %
      (p1.mean(), p1.min(), p1.max(), (T[2:, 0] - T[:-2, 1])[s].mean()))
tail_start = np.maximum(T[2:, 1], T[:-2, 2])  # the tail can start when its pass 1 is done and the queue's previous tail is done
tail_len = (T[2:, 2] - tail_start)[s]
wait_for_p1 = np.maximum(0, T[2:, 1] - T[:-2, 2])[s]      # tail queue idle, waiting for pass 1
wait_for_queue = np.maximum(0, T[:-2, 2] - T[2:, 1])[s]   # pass 1 done, the tail queue still busy with the previous tail
print("tail queue: a tail %.4f ms mean (min %.4f max %.4f); idle waiting for pass 1 %.4f per frame; a finished pass 1 waits for the queue %.4f per frame" %
      (tail_len.mean(), tail_len.min(), tail_len.max(), wait_for_p1.mean(), wait_for_queue.mean()))
lat = (T[:, 2] - T[:, 0])[s]
print("frame latency (bulk reaches it -> tail done) %.4f ms" % lat.mean())
