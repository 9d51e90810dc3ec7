"""PostProcessImage / AntiAliasImage (Engine3D/Renderer.cs:765-767, 819-978).

The reference holds no test or golden image for these two passes (its *xAA goldens use rayTraceSubPixelRes, which
samples different sub-pixel positions), so the numpy restatement in oracle/oracle_py.py is pinned by hand-computed
known answers below, and the HIP kernels are compared with it bit for bit."""
import numpy as np
import pytest

import softray_amd as sa
from softray_amd.renderer import Renderer, Style
from helpers import load_obj3ds, make_frame, orc

S = sa._lib


# ---------------------------------------------------------------- oracle known answers (CPU)
def test_oracle_style_known_answers():
    x = np.array([0xFF102030, 0x00000000, 0xFFFFFFFF, 0x80FF00FF, 0x00ABCDEF], dtype=np.uint32)
    assert orc.post_process(x, orc.STYLE_STANDARD).tolist() == x.tolist()
    # ((x & 0xffff) << 8) + ((x >> 16) & 0xff)
    assert orc.post_process(x, orc.STYLE_COLOR_SHUFFLE).tolist() == [0x00203010, 0, 0x00FFFFFF, 0x0000FFFF, 0x00CDEFAB]
    # x == bg ? bg : 0x00ffffff - x   (unchecked uint arithmetic wraps)
    assert orc.post_process(x, orc.STYLE_NEGATIVE, 0x00ABCDEF).tolist() == [
        (0x00FFFFFF - 0xFF102030) & 0xFFFFFFFF, 0x00FFFFFF, (0x00FFFFFF - 0xFFFFFFFF) & 0xFFFFFFFF,
        (0x00FFFFFF - 0x80FF00FF) & 0xFFFFFFFF, 0x00ABCDEF]
    assert orc.post_process(x, orc.STYLE_DEPTH_SMOOTH).tolist() == [0x00FFFFFF, 0, 0x00FFFFFF, 0x00808080, 0]
    assert orc.post_process(x, orc.STYLE_DEPTH_BANDED).tolist() == [255 * 111, 0, 255 * 111, 128 * 111, 0]
    with pytest.raises(ValueError):
        orc.post_process(x, 5)


def test_oracle_anti_alias_known_answers():
    # 2x2 -> 1 pixel: channel sums 10+20+31+40=101 -> 25 ; 0+0+0+255 -> 63 ; 255*4 -> 255 ; source alpha ignored
    src = np.array([0x000A00FF, 0x121400FF, 0xFF1F00FF, 0x7F28FFFF], dtype=np.uint32)
    assert orc.anti_alias(src, 1, 1, 2).tolist() == [0xFF193FFF]
    # resolution 1 = PackRgb(UnpackRgb(x)): alpha forced to 255
    assert orc.anti_alias(src, 2, 2, 1).tolist() == [0xFF0A00FF, 0xFF1400FF, 0xFF1F00FF, 0xFF28FFFF]
    # 2 destination pixels side by side, resolution 2: source is 4 wide, 2 high
    src = np.array([1, 3, 0x100, 0x300,
                    5, 8, 0x500, 0x900], dtype=np.uint32)
    assert orc.anti_alias(src, 2, 1, 2).tolist() == [0xFF000004, 0xFF000400]          # 17//4, 18//4


# ---------------------------------------------------------------- host logic (CPU, no device)
def test_passes_refuse_to_run_without_device():
    s = sa.GpuScene(device=-1)
    px = np.zeros(16, dtype=np.int32)
    with pytest.raises(sa.SoftrayError) as e:
        s.post_process(px, S.STYLE_NEGATIVE)
    assert e.value.code == S.SR_ERR_NO_DEVICE
    with pytest.raises(sa.SoftrayError) as e:
        s.anti_alias(px, 2, 2, 2)
    assert e.value.code == S.SR_ERR_NO_DEVICE
    with pytest.raises(sa.SoftrayError) as e:
        s.post_process(px, 5)                                    # Style.Normals
    assert e.value.code == S.SR_ERR_UNSUPPORTED
    with pytest.raises(sa.SoftrayError) as e:
        s.anti_alias(px, 2, 2, 0)
    assert e.value.code == S.SR_ERR_INVALID_ARG


def test_renderer_anti_alias_surface_bookkeeping():
    """AntiAliasResolution setter / SetRenderingSurface interplay, Renderer.cs:366-413, 593-626."""
    r = Renderer(device=-1)
    user = np.zeros(30 * 20, dtype=np.int32)
    r.SetRenderingSurface(30, 20, user)
    assert (r.RenderingSurfaceWidth, r.RenderingSurfaceHeight) == (30, 20) and r._pixels is user
    r.AntiAliasResolution = 3
    assert (r.RenderingSurfaceWidth, r.RenderingSurfaceHeight) == (90, 60)             # renders into the larger surface
    assert r._pixels.size == 90 * 60 and r._aaSurface[2] is user
    assert (r.rayTraceStartRow, r.rayTraceEndRow) == (0, 59)
    other = np.zeros(30 * 20, dtype=np.int32)
    r.SetRenderingSurface(30, 20, other)                                               # same size: only the buffer is swapped
    assert r._aaSurface[2] is other and r.RenderingSurfaceWidth == 90
    r.SetRenderingSurface(10, 10, np.zeros(100, dtype=np.int32))                       # new size under AA 3
    assert (r.RenderingSurfaceWidth, r.RenderingSurfaceHeight) == (30, 30) and r._aaSurface[:2] == (10, 10)
    r.AntiAliasResolution = 1                                                          # back to the caller's surface
    assert (r.RenderingSurfaceWidth, r.RenderingSurfaceHeight) == (10, 10) and r._aaSurface is None
    with pytest.raises(ValueError):
        r.AntiAliasResolution = 0
    r.Dispose()


# ---------------------------------------------------------------- GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("count", [0, 1, 3, 4, 5, 255, 1024, 1000003])
def test_post_process_matches_oracle(count):
    g = sa.GpuScene()
    rng = np.random.default_rng(count)
    base = rng.integers(0, 2 ** 32, size=count, dtype=np.uint64).astype(np.uint32)
    bg = 0x00123456
    if count > 4:
        base[::5] = bg                                           # pixels equal to BackgroundColor (Negative keeps them)
    for style in range(5):
        px = base.copy()
        g.post_process(px, style, bg)
        assert np.array_equal(px, orc.post_process(base, style, bg)), "style %d" % style


@pytest.mark.gpu
def test_post_process_device_unaligned_pointer_and_stream():
    import torch
    g = sa.GpuScene()
    rng = np.random.default_rng(7)
    base = rng.integers(0, 2 ** 32, size=4099, dtype=np.uint64).astype(np.uint32)
    t = torch.from_numpy(base.view(np.int32).copy()).cuda()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for off in (1, 2, 3):                                    # 4/8/12 bytes past a 16-byte boundary
            g.post_process_device(t.data_ptr() + 4 * off, 4099 - off - 2, S.STYLE_COLOR_SHUFFLE, 0, stream=st.cuda_stream)
    st.synchronize()
    want = base.copy()
    for off in (1, 2, 3):
        want[off:4097] = orc.post_process(want[off:4097], orc.STYLE_COLOR_SHUFFLE)
    assert np.array_equal(t.cpu().numpy().view(np.uint32), want)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,res", [(1, 1, 1), (1, 1, 2), (7, 5, 3), (64, 4, 2), (65, 5, 2), (100, 100, 4), (33, 17, 8), (512, 300, 2), (3, 2, 64)])
def test_anti_alias_matches_oracle(w, h, res):
    g = sa.GpuScene()
    rng = np.random.default_rng(w * 1000 + h * 10 + res)
    src = rng.integers(0, 2 ** 32, size=w * h * res * res, dtype=np.uint64).astype(np.uint32)
    got = g.anti_alias(src, w, h, res)
    assert np.array_equal(got, orc.anti_alias(src, w, h, res))


@pytest.mark.gpu
def test_anti_alias_full_size_properties():
    """4096^2 -> 1024^2 at resolution 4 (size-independent checks): a constant image is a fixed point, and the pass is
    monotone: averaging an image that is channel-wise >= another gives a result that is channel-wise >=."""
    import torch
    g = sa.GpuScene()
    n, res = 1024, 4
    a = torch.randint(0, 2 ** 31 - 1, (n * res * n * res,), dtype=torch.int32, device="cuda")
    hi = a | 0x00F0F0F0
    out_a = torch.empty(n * n, dtype=torch.int32, device="cuda")
    out_hi = torch.empty_like(out_a)
    g.anti_alias_device(a.data_ptr(), n, n, res, out_a.data_ptr())
    g.anti_alias_device(hi.data_ptr(), n, n, res, out_hi.data_ptr())
    const = torch.full_like(a, 0x00C86432)
    out_c = torch.empty_like(out_a)
    g.anti_alias_device(const.data_ptr(), n, n, res, out_c.data_ptr())
    torch.cuda.synchronize()
    assert bool((out_c == np.int32(np.uint32(0xFFC86432).view(np.int32))).all())
    for shift in (16, 8, 0):
        assert bool((((out_hi >> shift) & 0xFF) >= ((out_a >> shift) & 0xFF)).all())
    # spot-check one 64x64 destination block against the oracle
    blk = a.view(n * res, n * res)[: 64 * res, : 64 * res].contiguous().cpu().numpy()
    want = orc.anti_alias(blk.reshape(-1), 64, 64, res).reshape(64, 64)
    assert np.array_equal(out_a.view(n, n)[:64, :64].cpu().numpy().view(np.uint32), want)


@pytest.mark.gpu
def test_renderer_styles_and_anti_alias_end_to_end():
    """Renderer.Render() = raytrace, then PostProcessImage, then AntiAliasImage (Renderer.cs:746-767), against the
    same three steps done by the oracle."""
    import io, os
    from helpers import GOLDEN
    data = open(os.path.join(GOLDEN, "obj.3ds"), "rb").read()
    v9, argb, bmin, bmax = load_obj3ds()
    o = orc.Scene()
    o.set_triangles(v9, argb, bmin, bmax)
    o.build_tree()
    from softray_amd.renderer import Instance, Vector
    for style, aa in [(Style.Negative, 1), (Style.ColorShuffle, 2), (Style.Standard, 4), (Style.DepthSmooth, 2)]:
        with Renderer() as r:
            r.Load3dsModelFromStream(io.BytesIO(data))
            r.rayTrace = True
            r.rayTraceFocalBlur = False
            r.BackgroundColor = 0xFF00FF
            r.depthBuffer = True
            user = np.zeros(50 * 40, dtype=np.int32)
            r.SetRenderingSurface(50, 40, user)
            r.AntiAliasResolution = aa
            r.RenderStyle = style
            inst = Instance(r.Model, Position=Vector(0.0, 0.0, 1.0))
            r.Instances.append(inst)
            r.Render()
            frame = sa.Frame.from_buffer_copy(bytes(r.BuildFrame(inst)))
            assert (frame.width, frame.height) == (50 * aa, 40 * aa)
            of = orc.Frame.from_buffer_copy(bytes(frame))
            want, _ = o.render(of, threads=os.cpu_count())
            want = orc.post_process(want, style, 0xFF00FF)
            if aa > 1:
                want = orc.anti_alias(want, 50, 40, aa)
            assert np.array_equal(user.view(np.uint32), want), (style, aa)
