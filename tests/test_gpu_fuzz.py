"""Seeded random configurations, HIP path against the CPU oracle, bit for bit: scenes (triangle count, extent), surface sizes that
are no multiple of a tile, poses, light positions (inside and outside the scene's box), point / directional light, sample counts,
sub-pixel grids, mirror bounces, device- and host-built trees.  The fixed goldens pin the reference's own scenes; this sweep pins
the paths' edge handling (partial tiles, empty leaves' runs, lists that overflow, lights inside the box) on inputs nobody tuned for.

    python tests/test_gpu_fuzz.py 300 [first_seed]      # a longer sweep by hand (prints every case; stops at the first difference)
"""
import math
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (run by hand: python tests/test_gpu_fuzz.py)
import softray_amd as sa  # noqa: E402
from helpers import make_frame, orc, random_triangles  # noqa: E402

pytestmark = pytest.mark.gpu
NCPU = os.cpu_count() or 8


def case_of(seed):
    r = np.random.RandomState(seed)
    n = int(r.choice([40, 300, 1500, 6000, 25000]))
    extent = float(r.choice([0.05, 0.15, 0.4]) if n < 6000 else r.choice([0.02, 0.05, 0.15]))
    res_w = int(r.choice([33, 64, 100, 128, 200]))
    res_h = int(r.choice([17, 48, 90, 128, 160]))
    shadows = bool(r.rand() < 0.75)
    samples = int(r.choice([1, 5, 33, 64, 100, 128])) if shadows else 0
    spp = 2 if r.rand() < 0.2 else 1
    bounces = int(r.choice([0, 0, 0, 1, 3]))
    c = dict(seed=seed, n=n, extent=extent, res=(res_w, res_h), shadows=shadows, samples=samples, spp=spp, bounces=bounces,
             reflectivity=float(r.choice([0.3, 0.5, 1.0])), yaw=float(r.uniform(0, 360)), pitch=float(r.uniform(-80, 80)), roll=float(r.uniform(-30, 30)),
             depth=float(r.uniform(0.9, 1.7)), point_light=bool(r.rand() < 0.8), shading=bool(r.rand() < 0.9), specular=bool(r.rand() < 0.7),
             light_scale=float(r.choice([0.15, 0.5, 1.0, 1.0, 2.5])), light_turn=float(r.uniform(0, 2 * math.pi)), on_device=bool(r.rand() < 0.6),
             leaf=int(r.choice([1, 2, 4, 4, 8])), rng_seed=int(r.randint(1, 2 ** 31 - 1)))
    if bounces:
        c["shadows"], c["samples"], c["spp"] = False, 0, 1      # (the mirror-bounce extension is defined for frames without shadow rays)
    return c


def big_case_of(seed):
    """Frames large enough for the PERSISTENT form of the shaft walk once its grid is shrunk to one workgroup per CU (hook 831): the
    waves pull their tiles from the per-XCD lists, and from the second frame on in the longest-first order k_tile_order made."""
    c = case_of(seed)
    r = np.random.RandomState(seed + 77)
    c.update(n=int(r.choice([3000, 6000, 12000])), extent=float(r.choice([0.03, 0.05])), res=(int(r.choice([512, 640])), 384),
             shadows=True, samples=int(r.choice([17, 33])), spp=1, bounces=0, point_light=True, light_scale=float(r.choice([0.5, 1.0, 2.5])),
             pitch=float(r.uniform(-40, 40)), depth=float(r.uniform(1.0, 1.4)), persistent=True)
    return c


def frame_of(c):
    f = make_frame(c["res"][0], c["res"][1], shading=c["shading"], shadows=c["shadows"], sub_pixel_res=c["spp"], yaw_deg=c["yaw"], pitch_deg=c["pitch"],
                   roll_deg=c["roll"], depth=c["depth"], point_light=c["point_light"], specular=c["specular"], shadow_samples=c["samples"])
    f.random_seed = c["rng_seed"]
    # the light: the renderer's default position scaled (0.15: inside the model's box for most poses) and turned about the view axis
    ca, sn = math.cos(c["light_turn"]), math.sin(c["light_turn"])
    lx, ly, lz = f.light_pos_view[0], f.light_pos_view[1], f.light_pos_view[2]
    f.light_pos_view[0] = (ca * lx - sn * ly) * c["light_scale"]
    f.light_pos_view[1] = (sn * lx + ca * ly) * c["light_scale"]
    f.light_pos_view[2] = c["depth"] + (lz - 1.5) * c["light_scale"]
    dx, dy, dz = f.light_dir_view[0], f.light_dir_view[1], f.light_dir_view[2]
    f.light_dir_view[0], f.light_dir_view[1], f.light_dir_view[2] = ca * dx - sn * dy, sn * dx + ca * dy, dz
    f.max_bounces, f.reflectivity = c["bounces"], c["reflectivity"]
    if "light_view" in c:
        for i in range(3):
            f.light_pos_view[i] = c["light_view"][i]
    return f


def run_case(c):
    v9, argb, _ = random_triangles(c["n"], c["seed"] + 1000, space=1.0 - c["extent"], extent=c["extent"], origin=-0.5, mask_color=True)
    lo, hi = np.array([-0.5] * 3), np.array([0.5] * 3)
    o = orc.Scene()
    o.set_triangles(v9, argb, lo, hi)
    assert o.build_tree() == 0
    f = frame_of(c)
    want = np.zeros(c["res"][0] * c["res"][1], dtype=np.int32)
    o.render(f, threads=NCPU, out=want)
    g = sa.GpuScene(0)
    g.set_triangles(v9, argb, lo, hi)
    g.debug_set(sa._lib.DBG_BVH_LEAF, c["leaf"])
    g.build((sa.MODE_BVH,), on_device=c["on_device"])
    fs = sa.Frame.from_buffer_copy(bytes(f))
    fs.trace_mode = sa.MODE_BVH
    if c.get("persistent"):
        g.debug_set(sa._lib.DBG_KERNEL_SWITCH, 831)
        for _ in range(2):                                  # the frame after a frame walks the tiles in the order the last one's walk lengths give
            first, _ = g.render(fs)
            assert np.array_equal(np.asarray(first).view(np.uint32).ravel(), want.view(np.uint32).ravel()), "first frames (natural tile order) differ: %r" % (c,)
    got, _ = g.render(fs)
    return np.asarray(got).view(np.uint32).ravel(), want.view(np.uint32).ravel()


@pytest.mark.parametrize("seed", range(1, 33))
def test_random_configuration_equals_oracle(seed):
    c = case_of(seed)
    got, want = run_case(c)
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, "%d pixels differ (first %s) in %r" % (bad.size, bad[:5], c)


@pytest.mark.parametrize("seed", range(1, 5))
def test_random_large_frame_on_the_persistent_shaft_walk(seed):
    c = big_case_of(seed)
    got, want = run_case(c)
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, "%d pixels differ (first %s) in %r" % (bad.size, bad[:5], c)


@pytest.mark.parametrize("mask", range(8))
@pytest.mark.parametrize("below", [0, 7, 5])
def test_light_outside_the_box_on_every_subset_of_axes(mask, below):
    """The light-ordered node copy holds (near, far) planes on the axes where the light lies outside the scene's box (one instantiation of the
    shaft walk per subset); `below`: the axes on which it lies on the negative side."""
    c = case_of(4242 + mask)
    c.update(n=3000, extent=0.08, res=(200, 144), shadows=True, samples=33, spp=1, bounces=0, point_light=True, light_scale=1.0, light_turn=0.0,
             yaw=135.0, pitch=-22.0, roll=0.0, depth=1.3, on_device=True, leaf=4)
    model = [((-1.1 if (below >> a) & 1 else 1.2) if (mask >> a) & 1 else (-0.3 if (below >> a) & 1 else 0.25)) for a in range(3)]
    f0 = frame_of(c)
    t = [f0.transform[i] for i in range(12)]
    c["light_view"] = [t[4 * r] * model[0] + t[4 * r + 1] * model[1] + t[4 * r + 2] * model[2] + t[4 * r + 3] for r in range(3)]
    got, want = run_case(c)
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, "%d pixels differ (first %s), light at %r in model space" % (bad.size, bad[:5], model)


if __name__ == "__main__":
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    big = len(sys.argv) > 3 and sys.argv[3] == "big"            # python tests/test_gpu_fuzz.py 40 100 big: the large persistent-walk cases
    for seed in range(first, first + count):
        c = big_case_of(seed) if big else case_of(seed)
        got, want = run_case(c)
        bad = np.flatnonzero(got != want)
        print(seed, "ok" if bad.size == 0 else "DIFFERENT %d" % bad.size, c, flush=True)
        if bad.size:
            sys.exit(1)
    print("all", count, "equal")
