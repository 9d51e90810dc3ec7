"""ABI 5 additions on the GPU: sr_trace_rays_device (IntersectRay batches that stay in HBM), AxisAlignedBox as extra geometry, and the
native RCCL strip gather (sr_rccl_* / sr_set_gather).  The RCCL calls that need two GPUs are skip-gated, so the first multi-GPU box
that runs the suite validates them; what one GPU can exercise (communicator bootstrap, strip bookkeeping, de-interleave) runs here."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import softray_amd as sa
from helpers import ROOT, c1_spheres, load_obj3ds, make_frame, orc, random_triangles, unit_cube_scene
from test_oracle import TREE_BOX

pytestmark = pytest.mark.gpu
NCPU = os.cpu_count() or 8
NGPU = torch.cuda.device_count()


def as_sr(frame, mode):
    f = sa.Frame.from_buffer_copy(bytes(frame))
    f.trace_mode = mode
    return f


def test_trace_rays_device_equals_host_batches():
    """sr_trace_rays_device (device arrays, a stream, no host sync) == sr_trace_rays == the oracle, for the tree, brute force, the BVH
    and the root collection; NULL outputs are allowed."""
    v9, argb, rnd = random_triangles(1000, seed=12345)                  # RayIntersectTreeFromInside_Performance's tree (885 nodes)
    g = sa.GpuScene(0); o = orc.Scene()
    for s_ in (g, o):
        s_.set_triangles(v9, argb, *TREE_BOX)
    g.build((sa.MODE_REF_TREE, sa.MODE_BVH), 10, 5); assert o.build_tree(10, 5) == 0
    assert g.tree_stats() == o.tree_stats() == (10, 885, 443, 442)
    n = 100000
    u = rnd.NextDoubles(6 * n).reshape(n, 6)
    starts, dirs = u[:, :3] * 100.0, 2.0 * u[:, 3:] - 1.0
    dev = torch.device("cuda", 0)
    d_s, d_d = torch.from_numpy(starts).to(dev), torch.from_numpy(dirs).to(dev)
    st = torch.cuda.Stream(dev)
    for target, otarget in ((sa.MODE_REF_TREE, 1), (sa.MODE_BRUTE, 0), (sa.MODE_BVH, 3)):
        hit = torch.zeros(n, dtype=torch.uint8, device=dev); frac = torch.zeros(n, dtype=torch.float64, device=dev)
        pos = torch.zeros((n, 3), dtype=torch.float64, device=dev); nrm = torch.zeros((n, 3), dtype=torch.float64, device=dev)
        col = torch.zeros(n, dtype=torch.int32, device=dev); tri = torch.zeros(n, dtype=torch.int32, device=dev)
        with torch.cuda.stream(st):
            g.trace_device(target, n, d_s.data_ptr(), d_d.data_ptr(), hit.data_ptr(), frac.data_ptr(), pos.data_ptr(), nrm.data_ptr(), col.data_ptr(),
                           tri.data_ptr(), 0, st.cuda_stream)
            g.trace_device(target, n, d_s.data_ptr(), d_d.data_ptr(), hit.data_ptr(), 0, 0, 0, 0, 0, 0, st.cuda_stream)      # outputs may be NULL
        st.synchronize()
        b = o.trace(otarget, starts, dirs)
        a = g.trace(target, starts, dirs)
        got = dict(hit=hit.cpu().numpy(), ray_frac=frac.cpu().numpy(), pos=pos.cpu().numpy(), normal=nrm.cpu().numpy(),
                   color=col.cpu().numpy().view(np.uint32), tri_index=tri.cpu().numpy())
        for key in got:
            assert np.array_equal(got[key], b[key]) and np.array_equal(a[key], b[key]), (target, key)
        assert 0.2 < got["hit"].mean() < 0.3                            # the reference's hit-rate window (SpatialSubdivisionTests.cs:232)


def test_axis_aligned_box_as_extra_geometry():
    """AxisAlignedBox in ExtraGeometryToRaytrace (kind 4): ray batches (the reference's RayIntersectAABBPerformance rays, rays from
    inside, grazing rays) and frames (box + spheres + obj.3DS, with shadows: the box is an occluder too) == the oracle."""
    v9, argb, bmin, bmax = load_obj3ds()
    prims = [(4, 0, [-0.5, -0.5, -0.5, 0.5, 0.5, 0.5])] + c1_spheres(4) + [(4, 0, [0.3, -0.9, -0.2, 0.9, -0.6, 0.4])]
    g = sa.GpuScene(0); o = orc.Scene()
    for s_ in (g, o):
        s_.set_triangles(v9, argb, bmin, bmax)
        s_.set_extra(prims)
    g.build((sa.MODE_REF_TREE, sa.MODE_BVH)); assert o.build_tree() == 0
    rnd = orc.Random(12345)
    u = rnd.NextDoubles(6 * 100000).reshape(-1, 6)
    batches = [(np.stack([-0.3 + 0.6 * u[:, 0], -0.3 + 0.6 * u[:, 1], 0.5 + 0.5 * u[:, 2]], axis=1),
                np.stack([-0.5 + u[:, 3], -0.5 + u[:, 4], np.full(u.shape[0], -1.0)], axis=1)),
               (u[:, :3] - 0.5, 2.0 * u[:, 3:] - 1.0),                    # from inside the first box
               (4.0 * u[:, :3] - 2.0, (u[:, 3:] - 0.5) - (4.0 * u[:, :3] - 2.0) * 0.5)]
    for starts, dirs in batches:
        a = g.trace(sa.TARGET_ROOT | sa.MODE_BVH, starts, dirs, counters=True)
        b = o.trace(2, starts, dirs, counters=True)
        for key in ("hit", "tri_index", "color", "ray_frac", "pos", "normal"):
            assert np.array_equal(a[key], b[key]), key
    for kw in (dict(), dict(shadows=True), dict(sub_pixel_res=2, shadows=True)):
        f = make_frame(96, depth=3.0, **kw)
        want, _ = o.render(f, threads=NCPU)
        for mode in (sa.MODE_REF_TREE, sa.MODE_BRUTE, sa.MODE_BVH):
            for single in (False, True):
                fr = as_sr(f, mode)
                if single:
                    fr.flags |= sa._lib.F_SINGLE_KERNEL
                got, _ = g.render(fr)
                assert np.array_equal(got, want), (kw, mode, single)
    with pytest.raises(sa.SoftrayError):
        g.set_extra([(4, 0, [0.5, 0, 0, 0.5, 1, 1])])                     # min.x < max.x is required (AxisAlignedBox.cs:17)
    # through the Python host mirror
    from softray_amd.renderer import AxisAlignedBox, GeometryCollection, Vector
    with pytest.raises(ValueError):
        AxisAlignedBox(Vector(0, 0, 0), Vector(0, 1, 1))
    gc = GeometryCollection(); gc.Add(AxisAlignedBox(Vector(-0.5, -0.5, -0.5), Vector(0.5, 0.5, 0.5)))
    assert gc[0]._prim()[0] == 4


def test_rccl_gather_single_rank_and_refusals():
    """What one GPU can run of the native gather: ncclGetUniqueId / ncclCommInitRank with one rank, sr_rccl_render (= the rank's
    strips + the de-interleave; no peer to send to) against the plain frame, odd row ranges, and the argument checks."""
    v9, argb, bmin, bmax = unit_cube_scene(20000)
    g = sa.GpuScene(0)
    g.set_triangles(v9, argb, bmin, bmax)
    g.build((sa.MODE_BVH,))
    dev = torch.device("cuda", 0)
    f = as_sr(make_frame(150, 203, depth=1.5, shadows=True), sa.MODE_BVH)
    with pytest.raises(sa.SoftrayError):
        g.rccl_render(f, 0)                                             # before sr_rccl_init
    uid = sa.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    g.rccl_init(uid, 1, 0)
    st = torch.cuda.current_stream(dev)
    for kw in (dict(shadows=True), dict(start_row=5, end_row=190), dict(start_row=37, end_row=41), dict(sub_pixel_res=2)):
        f = as_sr(make_frame(150, 203, depth=1.5, **kw), sa.MODE_BVH)
        canvas = np.full(150 * 203, 0x12345678, dtype=np.int32)
        want, _ = g.render(f, out=canvas.copy())
        out = torch.from_numpy(canvas.copy()).to(dev)
        g.rccl_render(f, out.data_ptr(), st.cuda_stream)
        g.rccl_render(f, out.data_ptr(), st.cuda_stream)                # twice in a row, no host sync in between
        torch.cuda.synchronize(dev)
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want), kw
    with pytest.raises(sa.SoftrayError):
        g.rccl_render(as_sr(make_frame(64, depth=1.5, strips=(16, 2, 0)), sa.MODE_BVH), 0)      # the split is the library's
    with pytest.raises(sa.SoftrayError):
        g.rccl_init(uid, 2, 2)                                          # rank out of range
    with pytest.raises(sa.SoftrayError):
        g.set_gather(sa._lib.GATHER_RCCL)                               # not a multi-device scene
    multi = sa.GpuScene(devices=[0, 0])
    with pytest.raises(sa.SoftrayError):
        multi.set_gather(sa._lib.GATHER_RCCL)                           # RCCL refuses two ranks on one GPU: said up front
    multi.set_gather(sa._lib.GATHER_COPY)
    one = sa.GpuScene(devices=[0])
    one.set_gather(sa._lib.GATHER_RCCL)                                 # a one-device "node": nothing to exchange
    one.set_triangles(v9, argb, bmin, bmax); one.build((sa.MODE_BVH,))
    f = as_sr(make_frame(150, 203, depth=1.5, shadows=True), sa.MODE_BVH)
    out = torch.zeros(150 * 203, dtype=torch.int32, device=dev)
    one.render_device(f, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize(dev)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), g.render(f)[0])
    multi.close(); one.close(); g.close()


@pytest.mark.skipif(NGPU < 2, reason="needs two GPUs: grouped ncclSend / ncclRecv between the parts of a multi-device scene")
def test_in_library_rccl_gather_over_distinct_gpus():
    ndev = min(NGPU, 8)
    v9, argb, bmin, bmax = unit_cube_scene(20000)
    single = sa.GpuScene(0)
    single.set_triangles(v9, argb, bmin, bmax); single.build((sa.MODE_BVH,))
    multi = sa.GpuScene(devices=list(range(ndev)))
    multi.set_triangles(v9, argb, bmin, bmax); multi.build((sa.MODE_BVH,))
    multi.set_gather(sa._lib.GATHER_RCCL)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev)
    for kw in (dict(shadows=True), dict(start_row=5, end_row=500), dict(sub_pixel_res=2)):
        f = as_sr(make_frame(640, 515, depth=1.5, **kw), sa.MODE_BVH)
        canvas = np.full(640 * 515, 0x12345678, dtype=np.int32)
        want, _ = single.render(f, out=canvas.copy())
        out = torch.from_numpy(canvas.copy()).to(dev)
        multi.render_device(f, out.data_ptr(), st.cuda_stream)
        multi.render_device(f, out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize(dev)
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want), kw
        assert np.array_equal(multi.render(f, out=canvas.copy())[0], want)
    multi.close()


RANK_SCRIPT = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
rank, world, idfile, outfile = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
import numpy as np, torch
import softray_amd as sa
from helpers import make_frame, unit_cube_scene
torch.cuda.set_device(rank)
g = sa.GpuScene(rank)
g.set_triangles(*unit_cube_scene(20000)); g.build((sa.MODE_BVH,))
if rank == 0:
    open(idfile + ".tmp", "wb").write(sa.rccl_unique_id()); os.replace(idfile + ".tmp", idfile)
t0 = time.time()
while not os.path.exists(idfile):
    assert time.time() - t0 < 120; time.sleep(0.05)
g.rccl_init(open(idfile, "rb").read(), world, rank)
f = sa.Frame.from_buffer_copy(bytes(make_frame(640, 515, depth=1.5, shadows=True))); f.trace_mode = sa.MODE_BVH
dev = torch.device("cuda", rank)
out = torch.zeros(640 * 515, dtype=torch.int32, device=dev) if rank == 0 else None
for _ in range(3):
    g.rccl_render(f, out.data_ptr() if rank == 0 else 0, torch.cuda.current_stream(dev).cuda_stream)
torch.cuda.synchronize(dev)
if rank == 0:
    want, _ = g.render(f)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want)
    open(outfile, "w").write("ok")
"""


@pytest.mark.skipif(NGPU < 2, reason="needs two GPUs: one process per GPU, communicator bootstrapped through a file, no PyTorch in the data path")
def test_process_per_gpu_rccl_render(tmp_path):
    world = min(NGPU, 4)
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), str(tmp_path / "id.bin"), str(tmp_path / "ok.txt")], env=env)
             for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=600) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert (tmp_path / "ok.txt").read_text() == "ok"
