"""The N>1 path on CPU: world_size-2 gloo.  Each rank produces its interleaved strips (here with the CPU
oracle standing in for the GPU renderer: this test is about ownership, padding, the gather and the
de-interleave, which are identical on RCCL), rank 0 must end up with exactly the single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import load_obj3ds, make_frame, orc

from softray_amd.distributed import StripGather, owned_rows


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, res_w, res_h, strip_rows, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        v9, argb, bmin, bmax = load_obj3ds()
        o = orc.Scene()
        o.set_triangles(v9, argb, bmin, bmax)
        assert o.build_tree() == 0
        sg = StripGather(res_w, res_h, strip_rows, world, rank, torch.device("cpu"))
        f = make_frame(res_w, res_h, shadows=True, strips=(strip_rows, world, rank))
        px, _ = o.render(f, threads=2)
        assert px.size == sg.counts[rank]
        sg.local[: px.size] = torch.from_numpy(px.view(np.int32).copy())
        full = sg.exchange()
        # the pipelined form bench.py uses: two buffer sets, gathers in flight while the next frame is produced, finish() later
        sets = [StripGather(res_w, res_h, strip_rows, world, rank, torch.device("cpu")) for _ in range(2)]
        works = []
        for k, s2 in enumerate(sets):
            s2.local[: px.size] = torch.from_numpy(px.view(np.int32).copy()) + k      # frame k = the frame with k added to every pixel
            works.append(s2.start())
        for k, s2 in enumerate(sets):
            works[k].wait()
            got = s2.finish()
            if rank == 0:
                assert torch.equal(got, full + k)
            else:
                assert got is None
        if rank == 0:
            np.save(out_path, full.numpy())
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("res_w,res_h,strip_rows", [(48, 40, 16), (40, 37, 4)])
def test_strip_gather_world2(tmp_path, res_w, res_h, strip_rows):
    world = 2
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), res_w, res_h, strip_rows, out), nprocs=world, join=True)
    got = np.load(out).view(np.uint32)
    v9, argb, bmin, bmax = load_obj3ds()
    o = orc.Scene()
    o.set_triangles(v9, argb, bmin, bmax)
    assert o.build_tree() == 0
    want, _ = o.render(make_frame(res_w, res_h, shadows=True), threads=4)
    assert np.array_equal(got.reshape(-1), want)


def test_owned_rows_partition():
    for h, sr_, w in ((4096, 16, 8), (37, 4, 3), (10, 16, 2)):
        seen = []
        for r in range(w):
            seen += owned_rows(h, sr_, w, r)
        assert sorted(seen) == list(range(h))
    assert owned_rows(64, 4, 3, 1, 10, 20) == [r for r in range(10, 21) if (r // 4) % 3 == 1]
