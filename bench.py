#!/usr/bin/env python3
"""bench.py — Mrays/s of the ray-trace hot path on MI355X (BASELINE.json metric).

A step = one frame of the workload rendered by the HIP path into HBM-resident buffers
(+ for N > 1 the RCCL gather of the per-rank stripe buffers to rank 0 and the
de-interleave into the frame).  Workload at N = 1: BASELINE.json configs[1] =
balls_low.p3f at 1024x1024, Whitted, MAX_DEPTH 4, BVH.  For N > 1 the frame grows so
that every GPU keeps 1024*1024 pixels of the same picture (weak scaling; N = 4 is the
2048x2048 of configs[3]); rows are dealt to ranks in 8-row stripes, round-robin.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|tri100k|cornell_pt]

Prints ONE JSON line on rank 0.  `roofline` and `cpu_baseline` follow DESIGN.md §Measurement.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "tri100k", "cornell_pt"],
                    help="cfg2 (default) = BASELINE configs[1]; cfg3/cfg4/cfg5 = configs[2]/[3]/[4] at their full sizes; "
                         "tri100k / cornell_pt = the same scenes at quick sizes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tile-order", default="cost", choices=["cost", "frame"],
                    help="p3d_config.tile_order: cost = tiles most-expensive-class first (schedule recorded by the first "
                         "launch, i.e. during warmup); frame = image order.  Scheduling only, same bits either way.")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (production).  gloo stages the gather through host memory: only for "
                         "rehearsing the N>1 code path with several ranks on ONE GPU")
    ap.add_argument("--gather-batch", type=int, default=8,
                    help="N > 1: frames rendered back to back into one buffer per collective (every frame still reaches "
                         "rank 0; a 0.15 ms frame cannot pay for a collective launch of its own)")
    ap.add_argument("--gather", default="u8", choices=["u8", "f32"],
                    help="N>1: what rank 0 collects per frame: the u8 image (img_Data, 3 B/px) or float RGB + hit IDs (16 B/px)")
    return ap.parse_args()


def workload_setup(name, n_gpus, p3d):
    """-> (scene path, config, base resolution, description)"""
    scenes = os.path.join(ROOT, "tests", "golden", "scenes")
    if name == "cfg1":  # BASELINE configs[0]: the reference's own CPU-runnable case, fixed 512x512
        return (os.path.join(scenes, "balls_low.p3f"), p3d.whitted_config(accel=p3d.ACCEL_NONE, max_depth=1), -512,
                "balls_low.p3f 512x512, Whitted MAX_DEPTH=1, no acceleration structure (BASELINE configs[0])")
    if name == "cfg2":
        return (os.path.join(scenes, "balls_low.p3f"), p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=4), 1024,
                "balls_low.p3f, Whitted MAX_DEPTH=4, BVH, no AA (BASELINE configs[1])")
    if name in ("tri100k", "cfg4"):
        sys.path.insert(0, os.path.join(ROOT, "scenes"))
        import make_tri100k
        path = "/tmp/p3d_tri100k_%d.p3f" % os.getuid()
        if int(os.environ.get("LOCAL_RANK", "0")) == 0 and not os.path.exists(path):
            make_tri100k.generate(path + ".tmp", res=1024)
            os.replace(path + ".tmp", path)
        while not os.path.exists(path):
            time.sleep(0.2)
        if name == "cfg4":  # BASELINE configs[3]: 2048x2048 whatever the GPU count (strong scaling)
            return (path, p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=6), -2048,
                    "100k random triangles 2048x2048, Whitted MAX_DEPTH=6, BVH (BASELINE configs[3])")
        return (path, p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=6), 1024,
                "100k random triangles, Whitted MAX_DEPTH=6, BVH (BASELINE configs[3] scene)")
    cornell = os.path.join(ROOT, "scenes", "cornell.p3f")
    if name == "cfg3":  # BASELINE configs[2]
        return (cornell, p3d.pathtrace_config(accel=p3d.ACCEL_BVH, spp_sqrt=16, max_depth=20), 1024,
                "cornell.p3f, path tracer 256 spp, MAX_DEPTH=20, BVH (BASELINE configs[2])")
    if name == "cfg5":  # BASELINE configs[4]: thin lens (aperture 10, focal 1), 4096 spp; fixed 1024x1024
        return (cornell, p3d.pathtrace_config(accel=p3d.ACCEL_BVH, spp_sqrt=64, max_depth=20, dof=1), -1024,
                "cornell.p3f aperture 10 focal 1, path tracer 4096 spp + DOF sampler, BVH (BASELINE configs[4])")
    return (cornell, p3d.pathtrace_config(accel=p3d.ACCEL_BVH, spp_sqrt=4, max_depth=20),
            512, "cornell.p3f, path tracer 16 spp, BVH (BASELINE configs[2] scene, reduced spp)")


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import p3d_amd as p3d

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                     "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the hot path has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    host_staged = world > 1 and args.backend == "gloo"

    scene_path, cfg, base, desc = workload_setup(args.workload, world, p3d)
    cfg.tile_order = p3d.TILE_ORDER_COST if args.tile_order == "cost" else p3d.TILE_ORDER_FRAME
    stripe_h = 8
    fixed = base < 0  # negative base: fixed frame size (strong scaling)
    base = abs(base)
    res = base if fixed else int(round(base * math.sqrt(world) / (stripe_h * world))) * stripe_h * world
    assert res % (stripe_h * world) == 0
    hs = p3d.HostScene(scene_path)
    hs.set_resolution(res, res)
    if args.workload == "cfg5":
        hs.set_lens(10.0, 1.0)
    dev = p3d.DeviceScene(hs, bvh=True, device=dev_index)
    tile = p3d.stripe_tile((res, res), rank, world, stripe_h) if world > 1 else dev.full_tile()
    n_local = tile.w * tile.h
    stream = torch.cuda.current_stream()

    # Every rank renders ALL outputs of its stripes into HBM: float RGB + hit IDs (one packed
    # allocation, 16 B/px) and the gamma-corrected u8 image (img_Data, 3 B/px).  Rank 0 collects one
    # of the two (--gather) for EVERY frame; B = --gather-batch consecutive frames share one
    # collective (their buffers are adjacent), double-buffered against the rendering of the next B.
    B = max(1, args.gather_batch) if world > 1 else 1
    packed_sz, u8_sz = p3d.packed_bytes(n_local), n_local * 3

    def new_bufs():
        return (torch.empty(B * packed_sz, dtype=torch.uint8, device="cuda"), torch.empty(B * u8_sz, dtype=torch.uint8, device="cuda"))
    bufs = [new_bufs(), new_bufs()]
    handles = [None, None]
    filled = [0, 0]       # frames rendered into each batch buffer since its last gather
    in_flight = [0, 0]    # frames of the batch a pending gather carries
    sent_seq = [0, 0]     # order in which the slots' collectives were launched
    last_frame = [None]   # (slot, index) of the newest frame assembled on rank 0
    pick = (lambda pair: pair[1]) if args.gather == "u8" else (lambda pair: pair[0])
    gdev = "cpu" if host_staged else "cuda"
    # (every rank keeps receive buffers: only rank 0 uses them unless the backend forces all_gather)
    gathered = ([[torch.empty(pick(bufs[0]).shape, dtype=torch.uint8, device=gdev) for _ in range(world)] for _ in range(2)]
                if world > 1 else None)
    frame8 = torch.empty((B, res, res, 3), dtype=torch.uint8, device=gdev) if rank == 0 and world > 1 else None
    frame_rgb = torch.empty((B, res, res, 3), dtype=torch.float32, device=gdev) if rank == 0 and world > 1 else None
    frame_hit = torch.empty((B, res, res), dtype=torch.int32, device=gdev) if rank == 0 and world > 1 else None

    def assemble(slot):
        if args.gather == "u8":
            p3d.assemble_frame8(gathered[slot], (res, res), world, stripe_h, frame8, batch=B)
        else:
            p3d.assemble_frame(gathered[slot], (res, res), world, stripe_h, frame_rgb, frame_hit, batch=B)
        last_frame[0] = in_flight[slot] - 1

    ev_pairs = []

    def render_into(pair, tile_, cfg_, stats=None, frame=0):
        packed, u8 = pair
        n_px = tile_.w * tile_.h
        base = packed.data_ptr() + frame * p3d.packed_bytes(n_px)
        dev.render_device(cfg_, tile_, d_rgb=base, d_hit=base + n_px * 12, d_rgb8=u8.data_ptr() + frame * n_px * 3,
                          stream=stream.cuda_stream, stats=stats)

    def finish(slot):  # the collective of this slot has to be complete before the buffer is reused
        if handles[slot] is not None:
            handles[slot].wait()
            if rank == 0:
                assemble(slot)
            handles[slot] = None

    def send(slot):
        if world > 1 and filled[slot]:
            payload = pick(bufs[slot]).cpu() if host_staged else pick(bufs[slot])
            handles[slot], _ = p3d.gather_frame(payload, (res, res), rank, world, stripe_h, 0, gathered[slot], async_op=True)
            in_flight[slot], filled[slot] = filled[slot], 0
            sent_seq[slot] = max(sent_seq) + 1

    def step(i, timed):
        slot, f = (i // B) & 1, i % B
        if f == 0:
            finish(slot)
        # kernel duration for the roofline: HIP events on the launch stream around every 8th timed
        # step (an event pair costs a few microseconds of stream time, comparable to 3 % of this frame)
        probe = timed and (i % 8 == 0)
        if probe:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
        render_into(bufs[slot], tile, cfg, frame=f)
        if probe:
            e1.record(stream)
            ev_pairs.append((e0, e1))
        filled[slot] += 1
        if f == B - 1:
            send(slot)

    def drain():  # a partly filled batch goes out as it is; then every collective is completed, oldest first
        for slot in (0, 1):
            if filled[slot]:
                finish(slot)
                send(slot)
        for slot in sorted((0, 1), key=lambda k: sent_seq[k]):
            finish(slot)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, False)
    drain()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    drain()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=gdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev_pairs) / max(1, len(ev_pairs))

    # counts (deterministic): one counted pass of this rank's tile outside the timed region
    cfg_counted = p3d.Config.from_buffer_copy(bytes(cfg))
    cfg_counted.collect_stats = 1
    st = p3d.Stats()
    render_into(bufs[0], tile, cfg_counted, stats=st)
    counts = torch.tensor([st.rays, st.algorithmic_bytes()], dtype=torch.float64, device="cpu" if host_staged else "cuda")
    if world > 1:
        dist.all_reduce(counts)
    rays_total, _ = counts.tolist()

    gather_check = None
    if world > 1 and rank == 0:
        # the last assembled frame must equal, bit for bit, what one GPU renders for the whole frame
        full = dev.full_tile()
        ref_pair = (torch.empty(p3d.packed_bytes(res * res), dtype=torch.uint8, device="cuda"),
                    torch.empty(res * res * 3, dtype=torch.uint8, device="cuda"))
        render_into(ref_pair, full, cfg)
        torch.cuda.synchronize()
        ref_packed, ref_u8 = ref_pair[0].to(gdev), ref_pair[1].to(gdev)
        k = last_frame[0]
        if args.gather == "u8":
            ok = bool(torch.equal(frame8[k].reshape(-1), ref_u8))
        else:
            ok = bool(torch.equal(frame_rgb[k].reshape(-1).view(torch.int32), ref_packed[: res * res * 12].view(torch.int32))
                      and torch.equal(frame_hit[k].reshape(-1), ref_packed[res * res * 12:].view(torch.int32)))
        gather_check = "ok" if ok else "MISMATCH"

    if rank == 0:
        value = rays_total * args.steps / dt / 1e6
        alg_bytes_launch = st.algorithmic_bytes()  # rank 0's launch
        achieved = alg_bytes_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic = None
        prof = os.path.join(ROOT, "profiles", "r01_pmc_hbm_bytes.json")
        if os.path.exists(prof) and args.workload == "cfg2" and world == 1:
            traffic = json.load(open(prof)).get("hbm_bytes_per_launch")
        valu = None
        pmc = os.path.join(ROOT, "profiles", "r01", "cfg2_pmc_summary.json")
        if os.path.exists(pmc) and args.workload == "cfg2" and world == 1 and kernel_ms > 0:
            # the ceiling that does apply: a wave64 VALU instruction holds its SIMD for
            # 4 cycles; 256 CUs x 4 SIMDs at 2.4 GHz; instruction count from the committed PMC pass
            insts = json.load(open(pmc))["SQ_INSTS_VALU"]["mean"]
            floor_ms = insts * 4 / (1024 * 2.4e9) * 1e3
            valu = {"wave_instructions": int(insts), "floor_ms": round(floor_ms, 4), "frac": round(floor_ms / kernel_ms, 4)}
        out = {
            "metric": "Mrays/s (primary+secondary)", "value": round(value, 1), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if fixed else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "resolution": [res, res], "rays_per_frame": int(rays_total),
                       "ray_definition": "one traversal query (closest-hit or any-hit)",
                       "outputs_per_rank": "float RGB + hit ID (16 B/px) + u8 image (3 B/px), all written to HBM",
                       "tile_order": args.tile_order,
                       "parallelism": "image rows in %d-row stripes, round-robin over %d GPU(s)%s"
                                      % (stripe_h, world,
                                         "; every frame's %s gathered to rank 0 over RCCL, %d frames per collective, "
                                         "double-buffered; gathered frame vs single-GPU frame: %s"
                                         % ("u8 image" if args.gather == "u8" else "float RGB + hit IDs", B, gather_check)
                                         if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                         "kernel": "whitted_kernel" if cfg.integrator == p3d.WHITTED or not cfg.antialiasing else "pt_kernel",
                         "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes_launch),
                         "valu_issue": valu,
                         "note": "algorithmic bytes (DESIGN.md) are served from LDS/L2, not HBM: the kernel is "
                                 "VALU/latency bound; traffic = measured HBM bytes"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload, scene_path, cfg, res)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(workload, scene_path, cfg, res):
    """The oracle (CPU port of the reference's algorithm, literal semantics) on this host:
    1 thread, as the reference is single-threaded.  Bounded sample: whole frames of the same
    workload for cfg2 (about 1 s each), a centred crop for the heavier ones."""
    from oracle import binding as ob
    sc = ob.Scene(scene_path)
    sc.set_resolution(res, res)
    if workload == "cfg5":
        sc.set_lens(10.0, 1.0)
    ocfg = ob.default_config(integrator=cfg.integrator, accel=cfg.accel, max_depth=cfg.max_depth,
                             spp_sqrt=cfg.spp_sqrt, antialiasing=cfg.antialiasing,
                             depth_of_field=cfg.depth_of_field, sample_disk=cfg.sample_disk,
                             soft_shadows=cfg.soft_shadows, sample_mode=cfg.sample_mode, seed=cfg.seed,
                             rng_mode=0, stack_mode=1, trace_zero_weight=1, math_mode=0, threads=1)
    if workload in ("cfg1", "cfg2"):
        x0 = y0 = 0
        w = h = res
        reps = 5
        sample = "%d whole frames of the same workload (%dx%d), best of %d" % (reps, res, res, reps)
    else:
        w = h = 256 if workload in ("tri100k", "cfg4", "cornell_pt") else (64 if workload == "cfg3" else 16)
        x0 = y0 = (res - w) // 2
        reps = 2
        sample = "centred %dx%d crop of the %dx%d frame, best of %d" % (w, h, res, res, reps)
    sc.render(ocfg, x0, y0, 8, 8)  # builds the BVH outside the timed region
    best, rays = None, 0
    for _ in range(reps):
        _, _, st = sc.render(ocfg, x0, y0, w, h)
        if best is None or st.seconds < best:
            best, rays = st.seconds, st.rays
    out = {"value": round(rays / best / 1e6, 3), "unit": "Mrays/s", "cores": 1, "kind": "port", "sample": sample}
    # all host cores (rows dealt round-robin), parallel semantics; informational
    n = os.cpu_count() or 1
    ocfg.stack_mode = 0
    ocfg.trace_zero_weight = 0
    ocfg.threads = n
    _, _, st = sc.render(ocfg, x0, y0, w, h)
    out["all_cores"] = {"value": round(st.rays / st.seconds / 1e6, 3), "cores": n}
    return out


if __name__ == "__main__":
    main()
