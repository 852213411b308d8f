"""N > 1 path on CPU: two processes, gloo.  The GPUs only change who fills the per-rank
stripe buffer (HIP kernel instead of the oracle) and which backend moves it (RCCL instead
of gloo); the partition (p3d.stripe_tile / stripe_rows), the packed buffer layout, the
single gather per frame and the de-interleave (p3d.assemble_frame) are the code bench.py
runs on the 8-GPU node."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES, STRIPE_H = 64, 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path, mode="gather"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import p3d_amd as p3d
    from oracle import binding as ob
    p3d._GATHER_MODE["mode"] = mode  # "all_gather" = the fallback collective
    sc = ob.Scene(os.path.join(ROOT, "tests", "golden", "scenes", "balls_low.p3f"))
    sc.set_resolution(RES, RES)
    cfg = ob.whitted_config(2, 3)
    tile = p3d.stripe_tile((RES, RES), rank, world, STRIPE_H)
    rows = p3d.stripe_rows((RES, RES), rank, world, STRIPE_H)
    assert len(rows) == tile.h == RES // world
    # fill this rank's packed buffer stripe by stripe (the oracle renders contiguous row blocks)
    n_local = tile.w * tile.h
    rgb = np.zeros((tile.h, RES, 3), np.float32)
    hit = np.zeros((tile.h, RES), np.int32)
    for s in range(tile.h // STRIPE_H):
        y0 = int(rows[s * STRIPE_H])
        r, h, _ = sc.render(cfg, 0, y0, RES, STRIPE_H)
        rgb[s * STRIPE_H:(s + 1) * STRIPE_H] = r
        hit[s * STRIPE_H:(s + 1) * STRIPE_H] = h
    buf = torch.from_numpy(np.concatenate([rgb.reshape(-1).view(np.uint8), hit.reshape(-1).view(np.uint8)]).copy())
    assert buf.numel() == p3d.packed_bytes(n_local)
    work, gathered = p3d.gather_frame(buf, (RES, RES), rank, world, STRIPE_H, dst=0, async_op=True)
    work.wait()
    # a batch of three frames in one collective (bench.py --gather-batch): frame 1 is the real one
    junk = torch.full_like(buf, 7)
    work3, gathered3 = p3d.gather_frame(torch.cat([junk, buf, junk]), (RES, RES), rank, world, STRIPE_H, dst=0, async_op=True)
    work3.wait()
    u8 = torch.from_numpy((rgb * 255).astype(np.uint8).reshape(-1).copy())
    work8, gathered8 = p3d.gather_frame(torch.cat([u8, u8 // 2]), (RES, RES), rank, world, STRIPE_H, dst=0, async_op=True)
    work8.wait()
    if rank == 0:
        frame_rgb, frame_hit = p3d.assemble_frame(gathered, (RES, RES), world, STRIPE_H)
        full_rgb, full_hit, _ = sc.render(cfg)
        ok = bool((frame_rgb.numpy().view(np.uint32) == full_rgb.view(np.uint32)).all()
                  and (frame_hit.numpy() == full_hit).all())
        b_rgb, b_hit = p3d.assemble_frame(gathered3, (RES, RES), world, STRIPE_H, batch=3)
        ok = ok and b_rgb.shape == (3, RES, RES, 3) and bool((b_rgb[1].numpy().view(np.uint32) == full_rgb.view(np.uint32)).all()
                                                             and (b_hit[1].numpy() == full_hit).all())
        f8 = p3d.assemble_frame8(gathered8, (RES, RES), world, STRIPE_H, batch=2)
        full8 = (full_rgb * 255).astype(np.uint8)
        ok = ok and bool((f8[0].numpy() == full8).all() and (f8[1].numpy() == full8 // 2).all())
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "gather"), (4, "gather"), (2, "all_gather")])
def test_stripe_gather_assemble_two_ranks(tmp_path, world, mode):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), out, mode), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_stripe_rows_cover_the_frame_exactly_once():
    sys.path.insert(0, ROOT)
    import p3d_amd as p3d
    for world, sh, ry in ((1, 8, 64), (2, 8, 64), (4, 16, 128), (8, 8, 2944), (8, 16, 2048)):
        seen = np.concatenate([p3d.stripe_rows((32, ry), r, world, sh) for r in range(world)])
        assert sorted(seen.tolist()) == list(range(ry))
    with pytest.raises(ValueError):
        p3d.stripe_tile((32, 100), 0, 8, 8)
