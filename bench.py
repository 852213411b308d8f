#!/usr/bin/env python3
"""bench.py — Mrays/s of the ray-trace hot path on MI355X (BASELINE.json metric).

A step = one frame of the workload rendered by the HIP path into HBM-resident buffers
(+ for N > 1 the RCCL gather of the per-rank stripe buffers to rank 0 and the
de-interleave into the frame).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ...] [--stack-mode literal|per_pixel]

Workloads
    N = 1 (default `cfg2`): BASELINE.json configs[1] = balls_low.p3f 1024x1024, Whitted MAX_DEPTH 4, BVH.
    N > 1 (default `cfg4`): BASELINE.json configs[3] = 100k random triangles 2048x2048, Whitted MAX_DEPTH 6, BVH,
        the SAME frame whatever N (strong scaling); rows dealt to ranks in 8-row stripes, round-robin; every frame
        ends with ONE collective that brings float RGB + hit IDs (16 B/px) to rank 0.
    `cfg2w`: the cfg2 picture weak-scaled (1024*1024 pixels per GPU), u8 image gathered, 8 frames per collective.

`value` is measured with the default p3d_config.stack_mode = P3D_STACK_LITERAL, whose frames are bit-identical to
the reference's serial pixel order (tests/test_gpu_parity.py); `per_pixel_stack` is the same frame with the stack
emptied at every primary sample (one launch, <= 1e-4 from the reference on sphere scenes, identical on the
100k-triangle scene).  Rays are counted as the reference's frame has them (one traversal query each).

On one GPU the timed loop keeps several frames in flight: frame i is rendered by device scene i % n into buffers of its own
— the caller-side way to overlap the latency-bound parts of a frame (its slowest tiles; the launches of the hit_stack
hand-off, which are short dependent launches) with the bulk of the others; every frame is rendered completely, all are
finished when the timed region ends.  Literal frames (the default): six scenes, pass 1 of frame i on one of two "bulk"
streams (i % 2) and everything behind it on one of two "tail" streams (p3d_scene_set_tail_stream), four HIP streams = the four
hardware queues; per-pixel frames (and `--tail-streams 0`): four scenes, whole frames on four streams.
`frame.kernel_ms` is ONE frame on its own (HIP events inside the library).

Prints ONE JSON line on rank 0.  `roofline` and `cpu_baseline` follow DESIGN.md "Measurement".
"""
import argparse
import hashlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CLOCK_HZ = 2.4e9       # MI355X_MICROARCH.md: max clock
SIMDS = 256 * 4        # 256 CUs x 4 SIMDs
VALU_CYCLES = 2        # a wave64 VALU instruction issues over 2 cycles on a SIMD-32 (MI355X_MICROARCH.md "Wave scheduling";
#                        4 cycles is what ONE wave alone sustains, and what round 1 mistook for the SIMD's rate)
HBM_PEAK_GBPS = 8000.0
LDS_PEAK_GBPS = 256 * 128 * CLOCK_HZ / 1e9  # 128 B per clock and CU
TIMED_REPEATS = 5      # the timed loop (exactly --steps steps, barrier + synchronize on both sides) is run this often: median counted, all reported


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None,
                    choices=["cfg1", "cfg2", "cfg2w", "cfg3", "cfg4", "cfg5", "tri100k", "cornell_pt"],
                    help="default: cfg2 (BASELINE configs[1]) on one GPU, cfg4 (configs[3], strong scaling) on several; "
                         "cfg3/cfg5 = configs[2]/[4] at their full sizes; cfg2w = cfg2 weak-scaled; tri100k / cornell_pt = quick sizes")
    ap.add_argument("--stack-mode", default="literal", choices=["literal", "per_pixel"],
                    help="p3d_config.stack_mode of the timed frames (include/p3d.h)")
    ap.add_argument("--frames-in-flight", type=int, default=None,
                    help="N = 1: consecutive frames go round-robin to this many device scenes (each with its own scratch and "
                         "hand-off records, p3d.h: different scenes are independent) on as many HIP streams, so that the "
                         "latency-bound tail of frame k (its slowest tiles, the redo launch of the literal hand-off) overlaps "
                         "the bulk of the following frames.  Default on one GPU: 6 with tail streams (literal frames), 4 without (1 = every frame waits for the one before); 2 per rank on several.")
    ap.add_argument("--tail-streams", type=int, default=2,
                    help="N = 1, literal stack mode: the hit_stack hand-off launches of every frame (everything behind pass 1) go to "
                         "this many streams of their own (p3d_scene_set_tail_stream) and pass 1 of all frames to --bulk-streams "
                         "streams, instead of whole frames on --frames-in-flight streams (then 6 frames in flight by default); 0 = off")
    ap.add_argument("--bulk-streams", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tile-order", default="cost", choices=["cost", "frame"],
                    help="p3d_config.tile_order: cost = tiles most-expensive-class first (schedule recorded by the first "
                         "launch, i.e. during warmup); frame = image order.  Scheduling only, same bits either way.")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (production).  gloo stages the gather through host memory: only for "
                         "rehearsing the N>1 code path with several ranks on ONE GPU")
    ap.add_argument("--gather-batch", type=int, default=None,
                    help="N > 1: frames rendered back to back into one buffer per collective (default 1; cfg2w: 8 — a "
                         "0.3 ms frame cannot pay for a collective launch of its own)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the N > 1 code path (process group, stripes, one collective per frame, assemble on rank 0) also "
                         "with a single rank: exercises the RCCL gather on a one-GPU box")
    ap.add_argument("--gather", default=None, choices=["u8", "f32"],
                    help="N>1: what rank 0 collects per frame: float RGB + hit IDs (16 B/px, default) or the u8 image (3 B/px, cfg2w default)")
    return ap.parse_args()


def workload_setup(name, p3d):
    """-> (scene path, config, base resolution (negative: fixed, strong scaling), description)"""
    scenes = os.path.join(ROOT, "tests", "golden", "scenes")
    if name == "cfg1":  # BASELINE configs[0]: the reference's own CPU-runnable case, fixed 512x512
        return (os.path.join(scenes, "balls_low.p3f"), p3d.whitted_config(accel=p3d.ACCEL_NONE, max_depth=1), -512,
                "balls_low.p3f 512x512, Whitted MAX_DEPTH=1, no acceleration structure (BASELINE configs[0])")
    if name == "cfg2":
        return (os.path.join(scenes, "balls_low.p3f"), p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=4), -1024,
                "balls_low.p3f 1024x1024, Whitted MAX_DEPTH=4, BVH, no AA (BASELINE configs[1])")
    if name == "cfg2w":
        return (os.path.join(scenes, "balls_low.p3f"), p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=4), 1024,
                "balls_low.p3f, Whitted MAX_DEPTH=4, BVH, no AA, 1024x1024 pixels per GPU (BASELINE configs[1] weak-scaled)")
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
        return (path, p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=6), -1024,
                "100k random triangles 1024x1024, Whitted MAX_DEPTH=6, BVH (BASELINE configs[3] scene)")
    cornell = os.path.join(ROOT, "scenes", "cornell.p3f")
    if name == "cfg3":  # BASELINE configs[2]
        return (cornell, p3d.pathtrace_config(accel=p3d.ACCEL_BVH, spp_sqrt=16, max_depth=20), -1024,
                "cornell.p3f 1024x1024, path tracer 256 spp, MAX_DEPTH=20, BVH (BASELINE configs[2])")
    if name == "cfg5":  # BASELINE configs[4]: thin lens (aperture 10, focal 1), 4096 spp; fixed 1024x1024
        return (cornell, p3d.pathtrace_config(accel=p3d.ACCEL_BVH, spp_sqrt=64, max_depth=20, dof=1), -1024,
                "cornell.p3f 1024x1024 aperture 10 focal 1, path tracer 4096 spp + DOF sampler, BVH (BASELINE configs[4])")
    return (cornell, p3d.pathtrace_config(accel=p3d.ACCEL_BVH, spp_sqrt=4, max_depth=20),
            -512, "cornell.p3f 512x512, path tracer 16 spp, BVH (BASELINE configs[2] scene, reduced spp)")


def frame_valu_instructions(summary, literal):
    """VALU wave-instructions of all launches of ONE frame of the profiled stack mode, from a PMC summary: the kernels that ran at
    least about once per traced frame (the one-off counting and schedule-building launches of the warm-up did not), without the
    launches of the OTHER stack mode (a profiled bench run also times its per-pixel / literal counterpart: `whitted_kernel`'s
    seventh template argument, LIT, is 0 for the per-pixel stack)."""
    import re
    dom_calls = summary["dominant"].get("calls", 1)
    total = 0.0
    for name, v in summary.get("kernels", {}).items():
        c = v.get("counters", {}).get("SQ_INSTS_VALU")
        if not c or v.get("calls", 0) < 0.5 * dom_calls or name.startswith("__amd_rocclr"):
            continue
        m = re.match(r"whitted_kernel<(.*)>", name)
        if m:
            lit = m.group(1).split(",")[6].strip()
            if (lit == "0") == bool(literal):
                continue
        elif not literal and not name.startswith("pt_kernel"):
            continue
        total += c["mean"] * v["calls"] / dom_calls
    return total


def kernel_source_hash():
    """SHA-256 over the device sources: a PMC summary under profiles/ is only quoted for the kernels it was taken from."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "p3d-raytracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hpp", ".inc", ".hip")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_summary(workload, stack_mode):
    """profiles/r04/<workload>_<stack mode>_pmc_summary.json (profiles/r04_profile_recipe.sh) if it was collected
    from exactly these kernel sources, else None: stale counters are not quoted."""
    path = os.path.join(ROOT, "profiles", "r04", "%s_%s_pmc_summary.json" % (workload, stack_mode))
    if not os.path.exists(path):
        return None, "no PMC summary committed for this workload (%s)" % os.path.relpath(path, ROOT)
    s = json.load(open(path))
    if s.get("source_hash") != kernel_source_hash():
        return None, "PMC summary %s was collected from other kernel sources (%s, now %s)" % (
            os.path.relpath(path, ROOT), s.get("source_hash"), kernel_source_hash())
    return s, os.path.relpath(path, ROOT)


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
    dist_on = world > 1 or args.force_dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the hot path has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    host_staged = dist_on and args.backend == "gloo"

    workload = args.workload or ("cfg2" if not dist_on else "cfg4")
    gather = args.gather or ("u8" if workload == "cfg2w" else "f32")
    scene_path, cfg, base, desc = workload_setup(workload, p3d)
    cfg.tile_order = p3d.TILE_ORDER_COST if args.tile_order == "cost" else p3d.TILE_ORDER_FRAME
    cfg.stack_mode = p3d.STACK_LITERAL if args.stack_mode == "literal" else p3d.STACK_PER_PIXEL
    whitted = cfg.integrator == p3d.WHITTED or not cfg.antialiasing
    literal = cfg.stack_mode == p3d.STACK_LITERAL and whitted and cfg.accel == p3d.ACCEL_BVH
    stripe_h = 8
    fixed = base < 0  # negative base: fixed frame size (strong scaling)
    base = abs(base)
    res = base if fixed else int(round(base * math.sqrt(world) / (stripe_h * world))) * stripe_h * world
    assert res % (stripe_h * world) == 0
    hs = p3d.HostScene(scene_path)
    hs.set_resolution(res, res)
    if workload == "cfg5":
        hs.set_lens(10.0, 1.0)
    dev = p3d.DeviceScene(hs, bvh=True, device=dev_index)
    tile = p3d.stripe_tile((res, res), rank, world, stripe_h) if dist_on else dev.full_tile()
    n_local = tile.w * tile.h
    stream = torch.cuda.current_stream()
    # frames in flight (N = 1): frame i is rendered by scene i % nfl on stream i % nfl into output buffers of its own
    # N > 1: two (one per gather slot) - every rank renders frame k + 1 on its second device scene and stream while frame k's
    # slowest tiles finish and its gather runs.  A rank's stripes are a short launch, as long as its slowest waves whatever the
    # work in it: rank 0 of 8 of the configs[3] frame takes 3.45 ms alone and 2.09 ms per frame with two in flight
    # (an even share of the frame: 1.94 ms) - profiles/r04/experiments/README.md section 8.
    split_ok = not dist_on and args.stack_mode == "literal" and args.tail_streams > 0
    nfl = max(1, args.frames_in_flight if args.frames_in_flight is not None else ((6 if split_ok else 4) if not dist_on else 2))
    if dist_on:
        nfl = min(nfl, 2)  # (the gather is double-buffered: one scene and stream per slot)
    # (slot 0 stays on the default stream: HIP spreads streams over 4 hardware queues, the default stream has one to
    # itself and the pool streams share the other three — a fourth pool stream would queue behind the first one's launches)
    tail_streams = args.tail_streams if (split_ok and nfl > 1) else 0
    if tail_streams:
        # pass 1 of every frame on a few "bulk" streams (two such kernels fill the chip), the dependent hand-off launches behind
        # it on "tail" streams (p3d_scene_set_tail_stream): a frame's chain of short launches no longer holds up the next
        # frame's pass 1 on its stream.  Streams are created bulk first: HIP deals them to its hardware queues in that order.
        bulk = [stream] + [torch.cuda.Stream() for _ in range(max(1, args.bulk_streams) - 1)]
        tails = [torch.cuda.Stream() for _ in range(tail_streams)]
        scenes = [dev] + [p3d.DeviceScene(hs, bvh=True, device=dev_index) for _ in range(nfl - 1)]
        flight = [(scenes[k], bulk[k % len(bulk)]) for k in range(nfl)]
        for k, sc_ in enumerate(scenes):
            sc_.set_tail_stream(tails[k % len(tails)])
    else:
        flight = [(dev, stream)] + [(p3d.DeviceScene(hs, bvh=True, device=dev_index), torch.cuda.Stream()) for _ in range(nfl - 1)]

    # Every rank renders ALL outputs of its stripes into HBM: float RGB + hit IDs (one packed
    # allocation, 16 B/px) and the gamma-corrected u8 image (img_Data, 3 B/px).  Rank 0 collects one
    # of the two (--gather) for EVERY frame; B = --gather-batch consecutive frames share one
    # collective (their buffers are adjacent), double-buffered against the rendering of the next B.
    B = (args.gather_batch or (8 if workload == "cfg2w" else 1)) if dist_on else 1
    packed_sz, u8_sz = p3d.packed_bytes(n_local), n_local * 3

    def new_bufs():
        return (torch.empty(B * packed_sz, dtype=torch.uint8, device="cuda"), torch.empty(B * u8_sz, dtype=torch.uint8, device="cuda"))
    bufs = [new_bufs(), new_bufs()]
    handles = [None, None]
    filled = [0, 0]       # frames rendered into each batch buffer since its last gather
    in_flight = [0, 0]    # frames of the batch a pending gather carries
    sent_seq = [0, 0]     # order in which the slots' collectives were launched
    last_frame = [None]   # (slot, index) of the newest frame assembled on rank 0
    assembled = [0, 0]    # frames of the batch last assembled from each slot
    pick = (lambda pair: pair[1]) if gather == "u8" else (lambda pair: pair[0])
    gdev = "cpu" if host_staged else "cuda"
    # (every rank keeps receive buffers: only rank 0 uses them unless the backend forces all_gather)
    gathered = ([[torch.empty(pick(bufs[0]).shape, dtype=torch.uint8, device=gdev) for _ in range(world)] for _ in range(2)]
                if dist_on else None)
    # (one assembled frame per gather slot: the two slots' de-interleave copies run on different streams)
    asm = rank == 0 and dist_on
    frame8 = [torch.empty((B, res, res, 3), dtype=torch.uint8, device=gdev) if asm and gather == "u8" else None for _ in range(2)]
    frame_rgb = [torch.empty((B, res, res, 3), dtype=torch.float32, device=gdev) if asm and gather != "u8" else None for _ in range(2)]
    frame_hit = [torch.empty((B, res, res), dtype=torch.int32, device=gdev) if asm and gather != "u8" else None for _ in range(2)]

    def assemble(slot):
        if gather == "u8":
            p3d.assemble_frame8(gathered[slot], (res, res), world, stripe_h, frame8[slot], batch=B)
        else:
            p3d.assemble_frame(gathered[slot], (res, res), world, stripe_h, frame_rgb[slot], frame_hit[slot], batch=B)
        last_frame[0] = (slot, in_flight[slot] - 1)
        assembled[slot] = in_flight[slot]

    def render_into(pair, tile_, cfg_, stats=None, frame=0, scene=None, on=None):
        packed, u8 = pair
        n_px = tile_.w * tile_.h
        base_ = packed.data_ptr() + frame * p3d.packed_bytes(n_px)
        (scene or dev).render_device(cfg_, tile_, d_rgb=base_, d_hit=base_ + n_px * 12, d_rgb8=u8.data_ptr() + frame * n_px * 3,
                                     stream=(on or stream).cuda_stream, stats=stats)

    def slot_ctx(slot):  # N > 1: slot k renders on scene k and stream k; its collective and its de-interleave are ordered on that stream
        return torch.cuda.stream(flight[slot % nfl][1])

    def finish(slot):  # the collective of this slot has to be complete before the buffer is reused
        if handles[slot] is not None:
            with slot_ctx(slot):
                handles[slot].wait()
                if rank == 0:
                    assemble(slot)
            handles[slot] = None

    def send(slot):
        if dist_on and filled[slot]:
            with slot_ctx(slot):
                payload = pick(bufs[slot]).cpu() if host_staged else pick(bufs[slot])
                handles[slot], _ = p3d.gather_frame(payload, (res, res), rank, world, stripe_h, 0, gathered[slot], async_op=True)
            in_flight[slot], filled[slot] = filled[slot], 0
            sent_seq[slot] = max(sent_seq) + 1

    flight_bufs = [bufs[0]] + [new_bufs() for _ in range(nfl - 1)]

    def step(i, cfg_):
        if nfl > 1 and not dist_on:  # one GPU: nothing to gather; the frame goes to the scene, stream and buffers of its slot
            k = i % nfl
            render_into(flight_bufs[k], tile, cfg_, scene=flight[k][0], on=flight[k][1])
            return
        slot, f = (i // B) & 1, i % B
        if f == 0:
            finish(slot)
        render_into(bufs[slot], tile, cfg_, frame=f, scene=flight[slot % nfl][0], on=flight[slot % nfl][1])
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
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_loop(cfg_):
        """W untimed + exactly K timed steps, barrier + synchronize on both sides, max over ranks.  -> seconds"""
        for i in range(args.warmup):
            step(i, cfg_)
        drain()
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, cfg_)
        drain()
        barrier()
        dt = time.perf_counter() - t0
        if dist_on:
            tmax = torch.tensor([dt], dtype=torch.float64, device=gdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    # the first launch of a (scene, config, tile) key renders in image order and records the tile costs: time it on
    # its own (cold), before the warm-up that lets every later frame use the recorded schedule
    # (a 64-row tile first: loads the code objects of every kernel involved without touching the full tile's key)
    small = p3d.Tile(tile.x0, tile.y0, tile.w, min(tile.h, 64), tile.stripe_h, tile.stripe_stride)
    render_into(bufs[0], small, cfg, stats=p3d.Stats())
    cold = p3d.Stats()
    render_into(bufs[0], tile, cfg, stats=cold)
    for sc_, st_ in flight[1:]:  # the other slots record their tile schedules outside the timed region too
        render_into(bufs[0], tile, cfg, stats=p3d.Stats(), scene=sc_, on=st_)
    def timed_repeats(cfg_):
        """the timed loop up to TIMED_REPEATS times (fewer when one loop takes seconds: the heavy workloads), sorted"""
        first = timed_loop(cfg_)
        n = TIMED_REPEATS if first < 0.5 else (3 if first < 3.0 else 1)
        return sorted([first] + [timed_loop(cfg_) for _ in range(n - 1)])
    dts = timed_repeats(cfg)
    dt = dts[len(dts) // 2]

    # what the timed frames left in their slots must be, bit for bit, the frame one scene renders on its own
    flight_check = None
    if nfl > 1 and not dist_on:
        frames = [(b[0].clone(), b[1].clone()) for b in flight_bufs[:min(nfl, args.steps)]]
        render_into(bufs[1], tile, cfg)
        torch.cuda.synchronize()
        ok = all(torch.equal(f[0], bufs[1][0]) and torch.equal(f[1], bufs[1][1]) for f in frames)
        ok = ok and all(sc_.status() == 0 for sc_, _ in flight)
        flight_check = "ok" if ok else "MISMATCH"

    # Launch durations inside the frame, live: HIP events of the library on the launch stream (p3d_stats.kernel_ms /
    # pass1_ms / handoff_ms), 16 frames after the timed region, smallest and mean.
    probes = []
    n_probes = 16 if dt / args.steps < 0.05 else 3  # (frames of seconds: three)
    for _ in range(n_probes):
        st_ = p3d.Stats()
        render_into(bufs[0], tile, cfg, stats=st_)
        probes.append((st_.kernel_ms, st_.pass1_ms, st_.handoff_ms))
    kernel_ms = sum(p[0] for p in probes) / len(probes)
    pass1_ms = sum(p[1] for p in probes) / len(probes) if literal else kernel_ms
    handoff_ms = sum(p[2] for p in probes) / len(probes) if literal else 0.0

    # Rays of the frame as the reference counts them = the queries of the final frame.  The per-pixel-stack launch
    # traces exactly those (the literal launches trace some of them twice: that is overhead, not throughput).
    cfg_counted = p3d.Config.from_buffer_copy(bytes(cfg))
    cfg_counted.collect_stats = 1
    cfg_counted.stack_mode = p3d.STACK_PER_PIXEL
    st = p3d.Stats()
    render_into(bufs[0], tile, cfg_counted, stats=st)
    counts = torch.tensor([st.rays, st.algorithmic_bytes()], dtype=torch.float64, device="cpu" if host_staged else "cuda")
    if dist_on:
        dist.all_reduce(counts)
    rays_total, _ = counts.tolist()
    rays_total_rank0 = st.rays
    handoff = None
    if literal:
        cfg_counted.stack_mode = p3d.STACK_LITERAL
        hst = p3d.Stats()
        render_into(bufs[0], tile, cfg_counted, stats=hst)
        handoff = {"checked": int(hst.handoff_checked), "redone": int(hst.handoff_redone), "rounds": int(hst.handoff_rounds)}

    # the same frames with the stack emptied at every primary sample (one launch per frame): N = 1 only
    per_pixel = None
    if literal and not dist_on:
        cfg_pp = p3d.Config.from_buffer_copy(bytes(cfg))
        cfg_pp.stack_mode = p3d.STACK_PER_PIXEL
        pp_cold = p3d.Stats()
        fresh = p3d.DeviceScene(hs, bvh=True, device=dev_index)
        render_into(bufs[0], small, cfg_pp, stats=p3d.Stats(), scene=fresh)
        render_into(bufs[0], tile, cfg_pp, stats=pp_cold, scene=fresh)
        for sc_, st_ in flight:
            render_into(bufs[0], tile, cfg_pp, stats=p3d.Stats(), scene=sc_, on=st_)
        dts_pp = timed_repeats(cfg_pp)
        dt_pp = dts_pp[len(dts_pp) // 2]
        pk = []
        for _ in range(n_probes):
            st_ = p3d.Stats()
            render_into(bufs[0], tile, cfg_pp, stats=st_)
            pk.append(st_.kernel_ms)
        per_pixel = {"value": round(rays_total * args.steps / dt_pp / 1e6, 1), "ms_per_step": round(dt_pp / args.steps * 1e3, 4),
                     "ms_per_step_repeats": [round(d / args.steps * 1e3, 4) for d in dts_pp],
                     "kernel_ms": round(sum(pk) / len(pk), 4), "cold_kernel_ms": round(pp_cold.kernel_ms, 4),
                     "value_single_frame": round(rays_total / (sum(pk) / len(pk) * 1e-3) / 1e6, 1),
                     "note": "P3D_STACK_PER_PIXEL: hit IDs as the literal frame, colours within 1e-4 of it on this scene"}

    gather_check = None
    if dist_on and rank == 0:
        # the last assembled frame must equal, bit for bit, what one GPU renders for the whole frame
        full = dev.full_tile()
        ref_pair = (torch.empty(p3d.packed_bytes(res * res), dtype=torch.uint8, device="cuda"),
                    torch.empty(res * res * 3, dtype=torch.uint8, device="cuda"))
        render_into(ref_pair, full, cfg)
        torch.cuda.synchronize()
        ref_packed, ref_u8 = ref_pair[0].to(gdev), ref_pair[1].to(gdev)
        # ... every slot's last assembled frame (with two frames in flight: one per device scene and stream)
        ok = True
        for slot_ in range(2):
            if gather == "u8":
                ok = ok and all(bool(torch.equal(frame8[slot_][k].reshape(-1), ref_u8)) for k in range(assembled[slot_]))
            else:
                ok = ok and all(bool(torch.equal(frame_rgb[slot_][k].reshape(-1).view(torch.int32), ref_packed[: res * res * 12].view(torch.int32))
                                     and torch.equal(frame_hit[slot_][k].reshape(-1), ref_packed[res * res * 12:].view(torch.int32))) for k in range(assembled[slot_]))
        ok = ok and last_frame[0] is not None and all(sc_.status() == 0 for sc_, _ in flight)
        gather_check = "ok" if ok else "MISMATCH"
        # and what ONE GPU needs for the whole frame (outside the timed region, the other ranks idle): the N = 1 point of
        # THIS workload, so that the strong-scaling ratio can be read off one line (the driver's N = 1 run is cfg2)
        for _ in range(2):
            render_into(ref_pair, full, cfg)
        torch.cuda.synchronize()
        n1 = max(1, min(args.steps, 10))
        t1 = time.perf_counter()
        for _ in range(n1):
            render_into(ref_pair, full, cfg)
        torch.cuda.synchronize()
        single_gpu = {"ms_per_step": round((time.perf_counter() - t1) / n1 * 1e3, 4), "steps": n1,
                      "note": "the whole frame on rank 0's GPU alone, one frame at a time, no gather; value = rays_per_frame / this"}

    if rank == 0:
        value = rays_total * args.steps / dt / 1e6
        alg_bytes_launch = st.algorithmic_bytes()  # rank 0's launch
        dominant = "whitted_kernel" if whitted else "pt_kernel"
        dom_ms = pass1_ms if literal else kernel_ms
        alg_gbps = alg_bytes_launch / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # The nearest ceiling (DESIGN.md "Measurement"): the scene is LDS- or L2-resident, so HBM serves the
        # framebuffer only.  What the dominant kernel spends is instruction issue: a wave64 VALU instruction takes its
        # SIMD-32 for 2 cycles, 1024 SIMDs at 2.4 GHz = 1228.8 G wave-instructions/s.  Instruction and HBM byte counts come from the rocprofv3 PMC passes
        # of this workload committed under profiles/r04/ — quoted only if taken from exactly these kernel sources.
        summary, src = pmc_summary(workload, args.stack_mode) if not dist_on else (None, "PMC summaries are per single-GPU workload")
        staged = os.path.getsize(scene_path) < 64 * 1024  # (the packaged small scenes: staged in LDS by the kernels)
        roof = {"kernel": dominant, "kernel_ms": round(dom_ms, 4),
                "algorithmic": {"bytes_per_launch": int(alg_bytes_launch), "GBps": round(alg_gbps, 1),
                                "frac_vs_hbm": round(alg_gbps / HBM_PEAK_GBPS, 4),
                                "frac_vs_lds": round(alg_gbps / LDS_PEAK_GBPS, 4) if staged else None,
                                "note": "SURVEY.md 8(d) definition: bytes the algorithm looks at / dominant kernel's duration.  They are served by "
                                        "LDS (staged scenes) or L2 / Infinity Cache, not by HBM: frac_vs_hbm is NOT a bound when > 1 (nothing is "
                                        "skipped: counters equal the oracle's query by query); frac_vs_lds prices them against the LDS "
                                        "bandwidth of the chip (128 B/clk/CU); measured HBM traffic is under `traffic` / `hbm`"}}
        if summary:
            k = summary["dominant"]
            insts = k["SQ_INSTS_VALU"]
            achieved = insts / (dom_ms * 1e-3) / 1e9  # G wave-instructions / s
            peak = SIMDS * CLOCK_HZ / VALU_CYCLES / 1e9
            traffic = k.get("hbm_bytes")
            # every launch of a frame (hand-off launches included), not only the dominant one: the kernels of the summary that ran
            # about once per traced frame (the one-off counting and schedule-building launches of the warm-up do not)
            frame_insts = frame_valu_instructions(summary, literal) or insts
            roof.update({"bound": "valu_issue", "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "Gwave-instr/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "hbm": {"achieved_GBps": round(traffic / (dom_ms * 1e-3) / 1e9, 1) if traffic else None, "peak_GBps": HBM_PEAK_GBPS,
                                 "frac": round(traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None},
                         "wave_instructions_valu": int(insts), "lane_utilisation": k.get("lane_utilisation"),
                         "rocprof_avg_ms": k.get("avg_ms"), "source": src,
                         "scope": "the dominant kernel of ONE frame on its own (kernel_ms live from HIP events, rocprof_avg_ms from "
                                  "the committed trace of `bench.py --frames-in-flight 1`)",
                         "timed_loop_frac_lower_bound": round(insts / (dt / args.steps) / 1e9 / peak, 4) if not dist_on else None,
                         "timed_loop_frac": round(frame_insts / (dt / args.steps) / 1e9 / peak, 4) if not dist_on else None,
                         "timed_loop_note": "VALU wave-instructions of ALL launches of one frame / ms_per_step / peak: what the chip issues "
                                            "while `frames_in_flight` frames overlap (the lower bound counts the dominant kernel only)"})
        else:
            roof.update({"bound": "valu_issue", "achieved": None, "peak": round(SIMDS * CLOCK_HZ / VALU_CYCLES / 1e9, 1), "unit": "Gwave-instr/s",
                         "frac": None, "traffic": None, "source": src})
        out = {
            "metric": "Mrays/s (primary+secondary)", "value": round(value, 1), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "ms_per_step_repeats": [round(d / args.steps * 1e3, 4) for d in dts],
            "value_single_frame": round(rays_total_rank0 / (kernel_ms * 1e-3) / 1e6, 1) if not dist_on else None,
            "latency_ms_single_frame": round(kernel_ms, 4),
            "scaling": "strong" if fixed else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "resolution": [res, res], "rays_per_frame": int(rays_total),
                       "ray_definition": "one traversal query (closest-hit or any-hit) of the frame as the reference renders it",
                       "stack_mode": args.stack_mode if whitted and cfg.accel == p3d.ACCEL_BVH else "n/a (no stack survives a query here)",
                       "outputs_per_rank": "float RGB + hit ID (16 B/px) + u8 image (3 B/px), all written to HBM",
                       "tile_order": args.tile_order,
                       "frames_in_flight": nfl,
                       "streams": ("%d bulk (pass 1) + %d tail (hand-off launches)" % (len(bulk), len(tails))) if tail_streams else ("%d, whole frames" % nfl),
                       "frames_in_flight_check": ("each slot's last timed frame vs the frame one scene renders alone, all outputs: %s" % flight_check) if flight_check else None,
                       "parallelism": "image rows in %d-row stripes, round-robin over %d GPU(s)%s"
                                      % (stripe_h, world,
                                         "; every frame's %s gathered to rank 0 over %s, %d frame(s) per collective, "
                                         "double-buffered, %d frame(s) in flight per rank (one device scene and stream per gather slot); "
                                         "gathered frame vs single-GPU frame: %s"
                                         % ("u8 image" if gather == "u8" else "float RGB + hit IDs", "RCCL" if args.backend == "nccl" else "gloo (host-staged)", B, nfl, gather_check)
                                         if dist_on else "")},
            "value_note": "value = throughput of the timed loop (median of %d repeats of exactly --steps steps) with `frames_in_flight` frames "
                          "overlapping on as many device scenes (streams: config.streams); value_single_frame / latency_ms_single_frame = ONE frame rendered "
                          "alone (HIP events inside the library, mean of %d frames)" % (len(dts), n_probes),
            "frame": {"kernel_ms": round(kernel_ms, 4), "pass1_ms": round(pass1_ms, 4), "handoff_ms": round(handoff_ms, 4),
                      "cold_kernel_ms": round(cold.kernel_ms, 4), "handoff": handoff,
                      "note": "ONE frame on its own: HIP events of the library on the launch stream (ms_per_step is the rate of the "
                              "timed loop, with frames_in_flight frames overlapping); cold = first launch of this scene and config "
                              "(image order, tile costs recorded), the timed frames use the recorded tile schedule"},
            "roofline": roof,
        }
        if per_pixel:
            out["per_pixel_stack"] = per_pixel
        if dist_on:
            single_gpu["value"] = round(rays_total / (single_gpu["ms_per_step"] * 1e-3) / 1e6, 1)
            out["single_gpu_same_workload"] = single_gpu
        if not args.no_cpu_baseline and not dist_on:
            out["cpu_baseline"] = cpu_baseline(workload, scene_path, cfg, res)
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


def one_socket_cpus():
    """Logical CPUs of one socket this process may run on, one of them per physical core, and how many sockets the host has."""
    allowed = sorted(os.sched_getaffinity(0))
    by_pkg = {}
    for c in allowed:
        try:
            pkg = int(open("/sys/devices/system/cpu/cpu%d/topology/physical_package_id" % c).read())
        except OSError:
            pkg = 0
        by_pkg.setdefault(pkg, []).append(c)
    pkg = min(by_pkg)
    first_of_core = {}
    for c in by_pkg[pkg]:
        try:
            core = int(open("/sys/devices/system/cpu/cpu%d/topology/core_id" % c).read())
        except OSError:
            core = c
        first_of_core.setdefault(core, c)
    return pkg, by_pkg[pkg], len(by_pkg), sorted(first_of_core.values())


def cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max / v1 cfs quota), None = unlimited or unknown.
    A box that hands this process 16 CPUs' worth cannot show a 64-core socket scaling, whatever the code does."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except (OSError, ValueError):
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(workload, scene_path, cfg, res):
    """The oracle (CPU port of the reference's algorithm) on this host, bounded samples of the same workload:
      * 1 thread, reference-literal semantics (the reference is single-threaded): repeated until 10 s of CPU work;
      * every logical CPU of ONE socket, per-pixel stack (the serial stack cannot be threaded), rows handed out
        dynamically, three runs of at least a second each (the tile repeated inside one thread pool), all three reported.
    Whole frames for cfg1 / cfg2, a centred crop for the heavier workloads."""
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
        what = "whole frames of the same workload (%dx%d)" % (res, res)
    else:
        w = h = 256 if workload in ("tri100k", "cfg4", "cornell_pt") else (64 if workload == "cfg3" else 16)
        x0 = y0 = (res - w) // 2
        what = "renders of the centred %dx%d crop of the %dx%d frame" % (w, h, res, res)
    sc.render(ocfg, x0, y0, 8, 8)  # builds the BVH outside the timed region
    runs, rays, spent = [], 0, 0.0
    while len(runs) < 3 or (spent < 10.0 and len(runs) < 64):
        _, _, st = sc.render(ocfg, x0, y0, w, h)
        runs.append(st.seconds)
        spent += st.seconds
        rays = st.rays
    runs.sort()
    best, median = runs[0], runs[len(runs) // 2]
    out = {"value": round(rays / median / 1e6, 3), "unit": "Mrays/s", "cores": 1, "kind": "port",
           "sample": "%d %s, %.1f s of CPU work, median counted (fastest run: %.3f Mrays/s)" % (len(runs), what, spent, rays / best / 1e6),
           "semantics": "reference-literal: one hit_stack for the frame, zero-weight reflection rays traced", "cpu": cpu_model()}
    pkg, cpus, n_pkgs, core_cpus = one_socket_cpus()
    old = os.sched_getaffinity(0)
    ocfg.stack_mode = 0
    ocfg.trace_zero_weight = 0

    def timed(pin, threads, runs=3):
        """`runs` renders of >= 1.5 s each by one pool of `threads` threads confined to the CPUs `pin` -> sorted Mrays/s"""
        os.sched_setaffinity(0, pin)
        try:
            ocfg.threads = threads
            probe = sc.render_repeat(ocfg, 1, x0, y0, w, h)          # also warms the threads' code and data
            repeat = max(1, int(math.ceil(1.5 / max(probe.seconds, 1e-4))))
            vals = []
            for _ in range(runs):
                st = sc.render_repeat(ocfg, repeat, x0, y0, w, h)
                vals.append(st.rays / st.seconds / 1e6)
            return sorted(vals), repeat
        finally:
            os.sched_setaffinity(0, old)

    one, _ = timed(core_cpus[:1], 1, runs=1)                          # one thread, the SAME semantics: what the efficiency is measured against
    per_core, rep_c = timed(core_cpus, len(core_cpus))                # one thread per physical core
    per_cpu, rep_l = timed(cpus, len(cpus)) if len(cpus) > len(core_cpus) else (per_core, rep_c)  # one per logical CPU
    quota = cpu_quota()
    cands = [(per_core, len(core_cpus), rep_c, "one thread per physical core"), (per_cpu, len(cpus), rep_l, "one thread per logical CPU")]
    per_quota = None
    if quota and 1 <= int(quota) < len(core_cpus):  # the box grants fewer CPUs' worth of time than the socket has cores: as many threads as that
        per_quota, rep_q = timed(core_cpus[:int(quota)], int(quota))
        cands.append((per_quota, int(quota), rep_q, "as many threads as the container's CPU quota, one per physical core"))
    best, threads, rep_b, how = max(cands, key=lambda t: t[0][1])
    usable = min(len(core_cpus), quota) if quota else len(core_cpus)
    out["one_socket"] = {"value": round(best[1], 3), "unit": "Mrays/s", "runs": [round(v, 3) for v in best],
                         "spread": round((best[2] - best[0]) / best[1], 4),
                         "cores": len(core_cpus), "logical_cpus": len(cpus), "threads": threads, "pinning": how,
                         "socket": pkg, "sockets_on_host": n_pkgs,
                         "per_physical_core": [round(v, 3) for v in per_core], "per_logical_cpu": [round(v, 3) for v in per_cpu],
                         "per_quota_thread": [round(v, 3) for v in per_quota] if per_quota else None,
                         "single_thread_same_semantics": round(one[0], 3),
                         "speedup_over_one_thread": round(best[1] / one[0], 2),
                         "parallel_efficiency_per_core": round(best[1] / one[0] / len(core_cpus), 3),
                         "container_cpu_quota": quota,
                         "parallel_efficiency_vs_quota": round(best[1] / one[0] / usable, 3),
                         "semantics": "per-pixel stack (P3D_STACK_PER_PIXEL semantics; the serial stack cannot be threaded)",
                         "sample": "the same sample rendered %d times per run by one pool of pinned threads (cache-line-aligned per-thread "
                                   "state, rows handed out dynamically in blocks of 4), 3 runs of >= 1.5 s, median counted; "
                                   "container_cpu_quota = CPUs' worth of time the box grants this container (cgroup cpu.max), which caps "
                                   "the speed-up whatever the socket has" % rep_b}
    return out


if __name__ == "__main__":
    main()
