#!/usr/bin/env python3
"""bench.py -- headline benchmark of the raytrace hot path on MI355X.

Metric (BASELINE.json): Mrays/s = PRIMARY rays per second = W*H*subPixelRes^2 / t_frame, on the
1M-random-triangle scene (System.Random seed 12345) at 4096x4096, shading + the
reference's 100-sample dynamic soft shadows ("primary+shadow"), traced through the library's BVH.
A "step" is one full frame INCLUDING its way back to the host (SURVEY 8d / BASELINE.md 3: "pixel
read-back / RCCL gather included"): the scene, the BVH and the frame constants are resident in HBM
before the timed region; every step renders into one of two device surfaces and copies it into one
of two pinned host surfaces on a copy stream, so the read-back of frame k overlaps the rendering of
frame k + 1; the timed region ends when the last frame has arrived on the host.  The rate with the
frame left in HBM is reported beside it (`hbm_resident`).

    python bench.py --gpus N --steps K --warmup W

N > 1 (launched by torch.distributed.run, one rank per GPU): the frame is row-tiled in interleaved
16-row strips, every rank renders its strips into a compact device buffer and ONE RCCL gather over
xGMI collects the strips on rank 0 ("scaling": "strong" -- the frame is fixed), which reads it back.  The gather
is asynchronous (RCCL's stream), de-interleave and read-back follow it on the copy stream: frame k+1 renders meanwhile.
`--in-library` (single process): the same split behind the C ABI (sr_create_multi).
"""
import argparse
import copy
import json
import math
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# numpy / torch / the product library are imported by _imports(), AFTER the fan-out decision in main(): the parent of a plain
# `python bench.py --gpus N` only spawns torch.distributed.run as a child process and must never initialise the GPU itself
np = torch = dist = sa = R = StripGather = None


def _imports():
    global np, torch, dist, sa, R, StripGather
    import numpy as np_
    import torch as torch_
    import torch.distributed as dist_
    import softray_amd as sa_
    from softray_amd import renderer as R_
    from softray_amd.distributed import StripGather as SG_
    np, torch, dist, sa, R, StripGather = np_, torch_, dist_, sa_, R_, SG_


HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
VALU_PEAK_LANE_OPS = 256 * 4 * 16 * 2.4e9     # CUs x SIMDs x lanes per cycle x clock: 39.3 T lane-instructions / s (an FMA counts once)
S_NODE, S_TRI, S_SLAB, S_PIX = 64, 128, 64, 4  # bytes: BVH node, FP64 triangle record, fp32 TriSlab / CamCone record, pixel store


def profile_dir(args):
    """committed rocprofv3 summaries of this workload (the newest round that has them)"""
    if args.tris == 10000000:
        which = "bounces" if args.bounces > 0 else "shadows"
        cands = [os.path.join("r04_c5", which), os.path.join("r03_c5", "final_" + which), os.path.join("r03_c5", which)]
    else:
        cands = ["r04_final", "r03_final", "r02_final"]
    for c in cands:
        d = os.path.join(ROOT, "profiles", c)
        if os.path.exists(os.path.join(d, "sq_counters_summary.csv")):
            return d
    return os.path.join(ROOT, "profiles", cands[0])


def make_frame(args, strips=None):
    f = sa.Frame()
    f.width = f.height = args.res
    f.start_row, f.end_row = 0, args.res - 1
    f.sub_pixel_res = args.spp
    f.background_argb = 0xff00ff
    flags = sa.F_POINT_LIGHT | sa.F_SPECULAR | sa.F_SHADING
    if args.shadows > 0:
        flags |= sa.F_SHADOWS
        if getattr(args, "static_shadows", False):
            flags |= sa.F_STATIC_SHADOWS                              # rayTraceShadowsStatic: 128^3 cache, kept by the scene
    f.flags = flags
    f.random_seed = 1234567890
    f.shadow_samples = args.shadows if args.shadows > 0 else 0
    f.trace_mode = {"bvh": sa.MODE_BVH, "ref": sa.MODE_REF_TREE, "brute": sa.MODE_BRUTE}[args.mode]
    if strips:
        f.strip_rows, f.strip_count, f.strip_index = strips
    pos = [0.0, 0.0, args.depth]
    t, it = sa.instance_matrices(pos, 135.0 / 180.0 * math.pi, -22.0 / 180.0 * math.pi, 0.0)
    for i in range(12):
        f.transform[i] = t[i]
        f.inv_transform[i] = it[i]
    f.position_z = pos[2]
    f.fov_depth = sa.default_fov_depth()
    f.focal_depth = args.depth + 0.5
    f.focal_blur_strength = 10.0
    f.ambient, f.shininess = 0.1, 100.0
    d = R.Vector(-1, -1, 1)
    d.Normalise()
    lp = R.Vector(0.0, 0.0, 1.5) - d * 2                            # Renderer.cs:210-216
    for i, v in enumerate(d):
        f.light_dir_view[i] = v
    for i, v in enumerate(lp):
        f.light_pos_view[i] = v
    if getattr(args, "no_split", False):
        f.flags |= sa._lib.F_NO_SPLIT                                # one pipeline on the caller's stream (kernel timing)
    if args.bounces > 0:
        f.max_bounces, f.reflectivity = args.bounces, args.reflectivity
    if args.shadows == 1:                                            # hard-shadow variant: one sample, zero offset
        make_frame.zero = np.zeros(3)
        f.area_light_offsets = make_frame.zero.ctypes.data
    return f


def host_cores():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                            # honour a cgroup CPU quota (the GPU box gives each job a share)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(math.ceil(int(q) / int(per)))))
    except Exception:
        pass
    return cores


def cpu_baseline(args, v9, argb, bmin, bmax, budget_s=30.0):
    """The reference's CPU path = the C++ oracle (statement-level restatement; real C# cannot be built here), reference tree
    depth 15 / 25 per leaf, row-block threads like Renderer.cs:1659-1670, timed on a CENTRED SQUARE CROP of the SAME frame
    (rows and columns windowed, same rays; BASELINE.md 3: 256^2 of the 4096^2 frame, smaller when the host's cores would need
    more than the budget) at threads = all host cores (the headline `value`) and at threads = 4 (rayTraceConcurrency's default,
    Renderer.cs:82) on a 128^2 (or 64^2) crop.  The round-2 sample -- a band of whole centre rows -- is kept as `sample_rows`."""
    from oracle import oracle_py as orc                              # the only place bench.py touches oracle/
    cores = host_cores()
    o = orc.Scene()
    o.set_triangles(v9, argb, bmin, bmax)
    t0 = time.time()
    assert o.build_tree() == 0
    build_s = time.time() - t0
    f = orc.Frame.from_buffer_copy(bytes(make_frame(args)))
    f.trace_mode = orc.MODE_REF_TREE
    mid = args.res // 2
    scratch = np.zeros(args.res * args.res, dtype=np.int32)
    spp2 = args.spp * args.spp

    def crop(side, threads):
        side = min(side, args.res)
        a = mid - side // 2
        f.start_row, f.end_row = a, a + side - 1
        t0 = time.time()
        o.render(f, threads=threads, out=scratch, cols=(a, a + side))
        dt = time.time() - t0
        return side * side * spp2 / dt / 1e6, side * side * spp2, dt

    # all cores: 64^2 first (also the estimate for the larger crops), then the largest of 256^2 / 128^2 the budget allows
    v_all, rays_all, s_all = crop(64, cores)
    side_all = 64
    for side in (256, 128):
        if side <= args.res and s_all * (side / 64.0) ** 2 <= budget_s:
            v_all, rays_all, s_all = crop(side, cores)
            side_all = side
            break
    side_4 = 128 if s_all * (128.0 / side_all) ** 2 * cores / min(4, cores) <= budget_s else 64
    v_4, rays_4, s_4 = crop(side_4, min(4, cores))
    # the round-2 sample: whole rows through the centre of the frame (the most expensive rows: every pixel hit and shadowed)
    f.start_row, f.end_row = mid - 1, mid
    t0 = time.time()
    o.render(f, threads=cores, out=scratch)
    s_rows = time.time() - t0
    rays_rows = 2 * args.res * spp2
    # a C# toolchain on the GPU box would let the genuine reference be timed: probe, never assume
    csharp = [t for t in ("dotnet", "mono", "mcs", "csc") if shutil.which(t)]
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {"value": v_all, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "centred %dx%d crop (rows and columns windowed) of the same %dx%d frame (%d primary rays, %.1f s); C++ restatement of "
                      "the reference algorithm (reference tree 15/25, row-block threads, g++ -O2 -ffp-contract=off) -- real C# unavailable; "
                      "tree build %.1f s excluded" % (side_all, side_all, args.res, args.res, rays_all, s_all, build_s),
            "threads_4": {"value": v_4, "threads": min(4, cores), "sample": "centred %dx%d crop (%d primary rays, %.1f s)" % (side_4, side_4, rays_4, s_4),
                          "note": "rayTraceConcurrency's default (Renderer.cs:82)"},
            "sample_rows": {"value": rays_rows / s_rows / 1e6, "threads": cores,
                            "sample": "2 whole centre rows x %d cols (%d primary rays, %.1f s): the round-2 sample, biased low (every pixel of "
                                      "these rows is hit and shadowed; a third of the frame is background)" % (args.res, rays_rows, s_rows)},
            "cpu_model": cpu_model,
            "csharp_toolchain_on_this_box": csharp or None,
            "csharp_note": ("found %s: the genuine C# reference could be built here, but its sources do not travel to the GPU box" % csharp) if csharp
                           else "no dotnet / mono / mcs / csc on this box: the number is the C++ restatement of the reference algorithm"}


def committed_profile(PROFILE_DIR):
    """SQ counters / HBM traffic of this workload from the rocprofv3 passes committed under profiles/ (live PMC needs rocprofv3)."""
    out = {"issue": None, "traffic": {}}
    import csv
    spath = os.path.join(PROFILE_DIR, "sq_counters_summary.csv")
    if os.path.exists(spath):
        try:
            issue = {"source": os.path.relpath(spath, ROOT) + " (rocprofv3 --pmc SQ_*, separate passes, same workload)", "kernels": {}}
            for r in csv.DictReader(open(spath)):
                ns = float(r.get("_ns", 0) or 0)
                if ns <= 0:
                    continue
                lane_ops = float(r.get("SQ_THREAD_CYCLES_VALU", 0) or 0)
                issue["kernels"][r["kernel"]] = {
                    "valu_busy_frac": float(r["valu_busy_frac_at_2.4GHz_1024_SIMDs"]),
                    "active_lanes_per_valu_inst": float(r["active_lanes_per_valu_inst"]),
                    "useful_lane_ops_per_s": lane_ops / (ns * 1e-9),
                    "frac_of_valu_peak": lane_ops / (ns * 1e-9) / VALU_PEAK_LANE_OPS,
                }
            out["issue"] = issue
        except Exception:
            pass
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            out["traffic"] = json.load(open(tpath))
        except Exception:
            pass
    return out


# the reference's own published numbers: asserted throughput windows of its MSTest suite, AppVeyor build server, Release
REFERENCE_WINDOWS = {
    "ray_plane": (7.9, 10.8, "Mrays/s", "TriangleTests.cs:121-122"),
    "ray_triangle": (7.9, 12.9, "Mrays/s", "TriangleTests.cs:159-160"),
    "ray_sphere_from_inside": (4.9, 7.0, "Mrays/s", "TriangleTests.cs:200-201"),
    "ray_sphere_mostly_from_outside": (3.9, 6.5, "Mrays/s", "TriangleTests.cs:240-241"),
    "ray_sphere_randomly": (6.0, 9.5, "Mrays/s", "TriangleTests.cs:280-281"),
    "ray_aabb": (2.9, 4.1, "Mrays/s", "TriangleTests.cs:320-321"),
    "brute_force_1000_from_inside": (21.0, 29.0, "M ray-triangle/s", "SpatialSubdivisionTests.cs:158-159"),
    "tree_1000_from_inside": (60.0, 82.0, "M ray-triangle-equivalents/s", "SpatialSubdivisionTests.cs:204-205"),
    "tree_1000_mostly_from_outside": (25.0, 29.0, "M ray-triangle-equivalents/s", "SpatialSubdivisionTests.cs:248-249"),
}


def micro_benchmarks(args, dev):
    """The reference's only published figures, like for like: its per-primitive and per-tree IntersectRay micro-loops
    (TriangleTests.cs:100-345, SpatialSubdivisionTests.cs:148-282), with rays generated the way those tests generate them (System.Random;
    the tree tests' rays continue the seeded triangle stream), DEVICE-resident in and out (sr_trace_rays_device: the reference times
    IntersectRay, not a transfer), timed with HIP events on the launch stream.  Beside each: the reference's asserted window on its
    authors' build server and the oracle (the C++ restatement) on this host's cores, one thread, like the reference's loop."""
    g = sa.GpuScene(dev.index)
    stream = torch.cuda.current_stream(dev)
    n = args.micro_rays
    rng = lambda seed, k, skip=0: sa.net_random_doubles(seed, k, skip)

    def lerp3(u, lo, hi):                                             # MakeRandomVector(minX, maxX, minY, maxY, minZ, maxZ)
        lo, hi = np.asarray(lo, float), np.asarray(hi, float)
        return u * (hi - lo) + lo

    def rays_6(seed, count, skip=0):
        return rng(seed, 6 * count, skip).reshape(count, 6)

    empty = (np.zeros((0, 3, 3)), np.zeros(0, dtype=np.uint32), np.array([-1.0] * 3), np.array([1.0] * 3))
    # name -> (model, tree parameters, extra geometry, target, (starts, dirs), units of work per ray)
    u = rays_6(12345, n)
    ones = np.ones(3)
    cases = []
    cases.append(("ray_plane", empty, None, [(1, 0xffffffff, [0, 0, -100, 1, 1, 1])], sa.TARGET_ROOT | sa.MODE_BRUTE,
                  (u[:, :3] * ones, lerp3(u[:, 3:], [-1] * 3, [1] * 3)), 1))                      # Plane(forward * 100, (1, 1, 1)), :133-139
    cases.append(("ray_triangle", empty, None, [(2, 0xffffffff, [0, 0, 0, 1, 0, 0, 0, 1, 0])], sa.TARGET_ROOT | sa.MODE_BRUTE,
                  (u[:, :3] * ones, np.tile([0.0, 0.0, -1.0], (n, 1))), 1))                       # Triangle(origin, right, up), dir = forward, :171-178
    cases.append(("ray_sphere_from_inside", empty, None, [(0, 0xffffffff, [0.5, 0.5, 0.5, 1.0])], sa.TARGET_ROOT | sa.MODE_BRUTE,
                  (u[:, :3] * ones, lerp3(u[:, 3:], [-1] * 3, [1] * 3)), 1))                      # :211-219
    so = lerp3(u[:, :3], [-10] * 3, [10] * 3)
    cases.append(("ray_sphere_mostly_from_outside", empty, None, [(0, 0xffffffff, [0.5, 0.5, 0.5, 1.0])], sa.TARGET_ROOT | sa.MODE_BRUTE,
                  (so, u[:, 3:] * ones - so), 1))                                                  # :251-259
    cases.append(("ray_sphere_randomly", empty, None, [(0, 0xffffffff, [0.0, 0.0, 0.0, 1.0])], sa.TARGET_ROOT | sa.MODE_BRUTE,
                  (lerp3(u[:, :3], [-2] * 3, [2] * 3), lerp3(u[:, 3:], [-1] * 3, [1] * 3)), 1))   # :291-298
    cases.append(("ray_aabb", empty, None, [(4, 0xffffffff, [-0.5, -0.5, -0.5, 0.5, 0.5, 0.5])], sa.TARGET_ROOT | sa.MODE_BRUTE,
                  (lerp3(u[:, :3], [-0.3, -0.3, 0.5], [0.3, 0.3, 1.0]), lerp3(u[:, 3:], [-0.5, -0.5, -1.0], [0.5, 0.5, -1.0])), 1))   # :331-337
    # the seeded 1000-triangle soup (MakeRandomTriangles: space 100, extent 10, seed 12345); rays continue the SAME stream
    tv9, targb = sa.make_random_triangles(1000, 12345, space=100.0, extent=10.0)
    tbox = (np.array([0.0] * 3), np.array([110.0] * 3))                                           # GetBoundingBoxOfRandomTriangles
    ut = rays_6(12345, n, skip=10 * 1000)
    inside = (ut[:, :3] * 100.0, lerp3(ut[:, 3:], [-1] * 3, [1] * 3))                              # :181-183, :225-227
    st_out = ut[:, :3] * 1000.0
    outside = (st_out, ut[:, 3:] * 100.0 - st_out)                                                 # :269-271
    soup = (tv9, targb) + tbox
    cases.append(("brute_force_1000_from_inside", soup, None, [], sa.MODE_BRUTE, inside, 1000))
    cases.append(("tree_1000_from_inside", soup, (10, 5), [], sa.MODE_REF_TREE, inside, 1000))
    cases.append(("tree_1000_mostly_from_outside", soup, (10, 5), [], sa.MODE_REF_TREE, outside, 1000))
    cases.append(("own_bvh_1000_from_inside", soup, "bvh", [], sa.MODE_BVH, inside, 1000))

    from oracle import oracle_py as orc                              # cpu_baseline leg: the oracle timed beside the device
    results = {}
    hit = torch.zeros(n, dtype=torch.uint8, device=dev); frac = torch.zeros(n, dtype=torch.float64, device=dev)
    pos = torch.zeros((n, 3), dtype=torch.float64, device=dev); nrm = torch.zeros((n, 3), dtype=torch.float64, device=dev)
    col = torch.zeros(n, dtype=torch.int32, device=dev); tri = torch.zeros(n, dtype=torch.int32, device=dev)
    for name, model, tree, extra, target, (starts, dirs), work in cases:
        g.set_triangles(*model)
        g.set_extra(extra)
        if tree == "bvh":
            g.build((sa.MODE_BVH,))
        elif tree:
            g.build((sa.MODE_REF_TREE,), tree[0], tree[1])
        d_s = torch.from_numpy(np.ascontiguousarray(starts)).to(dev); d_d = torch.from_numpy(np.ascontiguousarray(dirs)).to(dev)
        call = lambda: g.trace_device(target, n, d_s.data_ptr(), d_d.data_ptr(), hit.data_ptr(), frac.data_ptr(), pos.data_ptr(), nrm.data_ptr(),
                                      col.data_ptr(), tri.data_ptr(), 0, stream.cuda_stream)
        call(); torch.cuda.synchronize(dev)
        best = float("inf")
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); call(); b.record(stream); b.synchronize()
            best = min(best, a.elapsed_time(b))
        h = hit.cpu().numpy()
        # CPU: the oracle, one thread, on a bounded sample of the same rays
        o = orc.Scene()
        o.set_triangles(*model)
        o.set_extra(extra)
        otarget = 2 if extra else {sa.MODE_BRUTE: 0, sa.MODE_REF_TREE: 1, sa.MODE_BVH: 3}[target & 0xff]
        if tree and tree != "bvh":
            assert o.build_tree(tree[0], tree[1]) == 0
        elif tree == "bvh":
            assert o.build_tree() == 0
        m = min(n, 1000000 if work == 1 else 20000)
        t0 = time.perf_counter()
        ob = o.trace(otarget, starts[:m], dirs[:m])
        cpu_s = time.perf_counter() - t0
        ref = REFERENCE_WINDOWS.get(name)
        results[name] = {
            "rays": n, "hit_rate": float(h.mean()), "device_ms": best, "device_Mrays_per_s": n / (best * 1e-3) / 1e6,
            "device_in_reference_units": n * work / (best * 1e-3) / 1e6,
            "reference_window": {"min": ref[0], "max": ref[1], "unit": ref[2], "source": ref[3], "hardware": "AppVeyor build server, Release, one thread"} if ref else None,
            "cpu_oracle": {"in_reference_units": m * work / cpu_s / 1e6, "rays": m, "seconds": cpu_s, "threads": 1,
                           "note": "C++ restatement incl. the ctypes call and the result arrays"},
            "first_rays_equal_oracle": bool(np.array_equal(h[:m], ob["hit"])) and bool(np.array_equal(frac[:m].cpu().numpy(), ob["ray_frac"])),
        }
    g.close()
    return {"metric": "IntersectRay micro-benchmarks of the reference (BASELINE.md 1), device-resident rays", "unit": "see entries", "n_gpus": 1,
            "data": "synthetic (System.Random seed 12345, generated as the reference's tests generate them)", "dtype": "f64",
            "micro": results,
            "note": "device_in_reference_units is in the unit of reference_window (ray-triangle products for the 1000-triangle cases, as the reference counts "
                    "them); one lane per ray, FP64, bit-identical to the oracle on the compared prefix"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--res", type=int, default=4096)
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--shadows", type=int, default=100, help="area-light samples per hit (100 = reference; 0 = primary only; 1 = hard shadow)")
    ap.add_argument("--spp", type=int, default=1, help="rayTraceSubPixelRes")
    ap.add_argument("--mode", default="bvh", choices=["bvh", "ref", "brute"])
    ap.add_argument("--depth", type=float, default=1.5)
    ap.add_argument("--no-split", action="store_true", help="SR_F_NO_SPLIT for every frame: one pipeline, no overlapping kernels")
    ap.add_argument("--static-shadows", action="store_true", help="rayTraceShadowsStatic (cache reset before every step: cold cache)")
    ap.add_argument("--bounces", type=int, default=0, help="config-5 extension: mirror bounces (parity unpinned; wavefront bounce pipeline on the own BVH)")
    ap.add_argument("--reflectivity", type=float, default=0.5)
    ap.add_argument("--extent", type=float, default=0.05, help="triangle extent of the synthetic soup (SURVEY 8d: 0.05 at 1M, 0.02 at 10M)")
    ap.add_argument("--strip-rows", type=int, default=16)
    ap.add_argument("--device-build", action="store_true", help="(the default since round 3) build the BVH on the GPU (LBVH)")
    ap.add_argument("--host-build", action="store_true", help="build the BVH with the host's binned-SAH builder instead of the device LBVH")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (kernel times, counters, primary-only, surface passes)")
    ap.add_argument("--exact-shadow-tests", action="store_true", help="k_shadow_test (every pair in FP64) instead of the fp32-classified k_shadow_cls")
    ap.add_argument("--dbg", action="append", default=[], metavar="KEY=VALUE", help="sr_debug_set hook, e.g. --dbg 1=16 (SR_DBG_ROUND_CAP0 = 16)")
    ap.add_argument("--in-library", action="store_true", help="single process: --gpus N devices behind sr_create_multi instead of one rank per GPU")
    ap.add_argument("--same-device", action="store_true", help="with --in-library: list device 0 N times (rehearsal on a one-GPU box)")
    ap.add_argument("--prelude-s", type=float, default=6.0, help="untimed seconds of continuous frames before the timed region (lets GPU-busy samplers see the GPU phase)")
    ap.add_argument("--verify", action="store_true", help="after timing: rank 0 re-renders the whole frame alone and compares it with the one that arrived on the host")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU rehearsals)")
    ap.add_argument("--no-verify", action="store_true", help="N > 1 verifies the gathered frame against a single-GPU render by default; this turns it off")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the ranks a plain `--gpus N` run spawns (0 = pick a free one)")
    ap.add_argument("--gather", default="torch", choices=["torch", "rccl"],
                    help="N > 1: the strip gather inside the timed region -- torch.distributed.gather, or the library's own grouped ncclSend / ncclRecv "
                         "(sr_rccl_render; with --in-library: sr_set_gather).  The other one is exercised once after the timed region either way")
    ap.add_argument("--micro", action="store_true", help="instead of the frame benchmark: the reference's IntersectRay micro-benchmarks (BASELINE.md 1), device-resident")
    ap.add_argument("--micro-rays", type=int, default=1 << 22)
    args = ap.parse_args()

    # ---- plain `python bench.py --gpus N` (no launcher): become the launcher.  BEFORE anything touches the GPU this process starts
    #      `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD (never exec: see the environment notes) and
    #      relays its output -- rank 0's JSON line included -- and its exit code ----
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.in_library:
        import socket
        import subprocess
        port = args.master_port
        if port <= 0:
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
            s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        sys.exit(subprocess.call(cmd, env=env))
    _imports()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU implementation")
    in_library = args.in_library and world == 1 and args.gpus > 1
    local_rank = local_rank % ndev            # (rehearsals with more ranks than GPUs share devices; the driver uses one rank per GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.micro:
        if world > 1 or args.gpus > 1:
            raise SystemExit("--micro is a one-GPU measurement")
        print(json.dumps(micro_benchmarks(args, dev)))
        return
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    # ---- scene resident in HBM (replicated on every GPU: 128 MB of records) ----
    v9, argb = sa.make_random_triangles(args.tris, 12345, space=1.0 - args.extent, extent=args.extent, origin=-0.5, opaque=True)
    bmin, bmax = np.array([-0.5] * 3), np.array([0.5] * 3)
    if in_library:
        devices = [0] * args.gpus if args.same_device else [d % ndev for d in range(args.gpus)]
        g = sa.GpuScene(devices=devices)
        if args.gather == "rccl":
            g.set_gather(sa._lib.GATHER_RCCL)                       # grouped ncclSend / ncclRecv between the parts (needs distinct devices)
    else:
        g = sa.GpuScene(local_rank)
    g.set_triangles(v9, argb, bmin, bmax)
    if args.exact_shadow_tests:
        g.debug_set(sa._lib.DBG_EXACT_SHADOW_TESTS, 1)
    for kv in args.dbg:
        k, v = kv.split("=")
        g.debug_set(int(k), int(v))
    t0 = time.time()
    g.build(({"bvh": sa.MODE_BVH, "ref": sa.MODE_REF_TREE}.get(args.mode),) if args.mode != "brute" else (),
            on_device=(False if args.host_build else None) if args.mode == "bvh" else None)
    build_s = time.time() - t0

    n_gpus = args.gpus if (world > 1 or in_library) else 1
    strips = (args.strip_rows, world, rank) if world > 1 else None
    frame = make_frame(args, strips)
    npix = g.pixel_count(frame)
    full_px = args.res * args.res
    stream = torch.cuda.current_stream(dev)
    copy_stream = torch.cuda.Stream(dev)
    sg = [StripGather(args.res, args.res, args.strip_rows, world, rank, dev) for _ in range(2)] if world > 1 else None
    if sg:
        assert sg[0].counts[rank] == npix
        surfaces = [s.full for s in sg]                                # rank 0: the gathered frames
    else:
        surfaces = [torch.empty(npix, dtype=torch.int32, device=dev) for _ in range(2)]
    host = [torch.empty(full_px, dtype=torch.int32).pin_memory() for _ in range(2)] if rank == 0 else None
    rendered = [torch.cuda.Event() for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]
    copied_once = [False, False]
    pending = [None, None]                                           # gather handles of the two buffer sets
    state = {"k": 0, "readback": True}

    full_frame = make_frame(args)                                    # the whole frame: sr_rccl_render deals the strips out itself
    native = {"ready": False}

    def native_init():
        """The library's own communicator (ncclCommInitRank through sr_rccl_init): rank 0 makes the id, the 128 bytes travel through
        torch.distributed's store -- plumbing; the pixels never touch PyTorch on this path."""
        if native["ready"]:
            return
        box = [sa.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        g.rccl_init(box[0], world, rank)
        native["ready"] = True

    use_native = world > 1 and args.gather == "rccl"
    if use_native:
        if args.strip_rows != 16:
            raise SystemExit("--gather rccl uses the library's 16-row strips")
        native_init()
        full_dev = [torch.zeros(full_px, dtype=torch.int32, device=dev) for _ in range(2)] if rank == 0 else [None, None]

    def step_native():
        k = state["k"] % 2
        state["k"] += 1
        if rank == 0 and copied_once[k]:
            stream.wait_event(copied[k])                             # surface k is free once its last read-back has finished
        g.rccl_render(full_frame, full_dev[k].data_ptr() if rank == 0 else 0, stream.cuda_stream)
        if rank == 0:
            rendered[k].record(stream)
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(rendered[k])
                if state["readback"]:
                    host[k].copy_(full_dev[k], non_blocking=True)
                copied[k].record(copy_stream)
            copied_once[k] = True

    def step():
        if use_native:
            return step_native()
        k = state["k"] % 2
        state["k"] += 1
        if args.static_shadows:
            g.reset_shadow_cache()                                   # every step generates the whole cache again
        if copied_once[k] and (state["readback"] or sg):
            stream.wait_event(copied[k])                             # surface k is free once its last read-back has finished
        if sg and pending[k] is not None:
            pending[k].wait()                                        # buffer set k: its previous gather has left / arrived (long ago)
        target = sg[k].local if sg else surfaces[k]
        g.render_device(frame, target.data_ptr(), stream.cuda_stream)
        if sg:
            # the gather is queued behind this frame's kernels and runs on RCCL's stream; the de-interleave and the read-back
            # follow it on the copy stream, so the compute stream goes straight on to the next frame (two buffer sets)
            pending[k] = sg[k].start()
            if rank == 0:
                with torch.cuda.stream(copy_stream):
                    pending[k].wait()
                    sg[k].finish()
                    if state["readback"]:
                        host[k].copy_(surfaces[k].view(-1), non_blocking=True)
                    copied[k].record(copy_stream)
                copied_once[k] = True
        elif rank == 0 and state["readback"]:
            rendered[k].record(stream)
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(rendered[k])
                host[k].copy_(surfaces[k].view(-1), non_blocking=True)   # pinned: a real asynchronous D2H
                copied[k].record(copy_stream)
            copied_once[k] = True

    def timed(steps):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)                                  # all streams of the device: the last frame is on the host
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if args.prelude_s > 0 and world == 1:                            # untimed: keeps the GPU visibly busy for a sampler
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < args.prelude_s:
            for _ in range(4):
                step()
            torch.cuda.synchronize(dev)
    elapsed = timed(args.steps)

    primary_rays = args.res * args.res * args.spp * args.spp
    value = primary_rays * args.steps / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    last_host_frame = host[(state["k"] - 1) % 2].numpy().copy() if rank == 0 else None

    out = None
    if rank == 0:
        out = {
            "metric": "Mrays/s (primary rays) at %dx%d, shading + %s, %s, frame read back to the host" % (
                args.res, args.res,
                "100-sample soft shadows" if args.shadows == 100 else ("no shadows" if args.shadows == 0 else "%d-sample shadows" % args.shadows),
                {"bvh": "own BVH", "ref": "reference tree", "brute": "brute force"}[args.mode]),
            "value": value, "unit": "Mrays/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d random triangles (System.Random seed 12345, extent %g, unit cube) + BVH, %dx%d, "
                                   "spp %d, shading + %d shadow samples/hit, pose yaw135/pitch-22/depth %.1f" % (
                                       args.tris, args.extent, args.res, args.res, args.spp * args.spp, args.shadows, args.depth),
                       "trace_mode": args.mode,
                       "parallelism": ("rows x%d (interleaved %d-row strips, %s)" % (
                           n_gpus, args.strip_rows, ("one process, sr_create_multi, gather: %s" % ("grouped ncclSend/ncclRecv" if args.gather == "rccl" else "peer copies")) if in_library
                           else ("one process per GPU + %s" % ("the library's grouped ncclSend/ncclRecv (sr_rccl_render)" if use_native else "torch.distributed.gather over RCCL")))) if n_gpus > 1 else "one GPU",
                       "readback": "every frame copied to pinned host memory inside the timed region (the copy of frame k overlaps frame k+1)"},
            "build_s": build_s,
        }
        if n_gpus > 1:
            out["frame_crc"] = int(np.bitwise_xor.reduce(last_host_frame.view(np.uint32)))

    if world > 1:
        # ---- what a reader needs to cross-check the N > 1 line against N = 1 (all untimed, after the timed region): who took
        #      part, every rank's own render time for its strips, and the three steps that follow the render on rank 0 ----
        for p in pending:
            if p is not None:
                p.wait()
        torch.cuda.synchronize(dev)
        reps = 3
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(reps):
            g.render_device(frame, sg[0].local.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(dev)
        my_ms = (time.perf_counter() - t1) / reps * 1e3
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "device": "cuda:%d" % local_rank, "name": props.name, "render_ms": my_ms,
                "rows": len(sg[0].rows[rank]), "pid": os.getpid()}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)

        def wall(fn):                                                # max over ranks of the wall time of fn() with the device drained
            dist.barrier()
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            fn()
            torch.cuda.synchronize(dev)
            v = torch.tensor([(time.perf_counter() - t) * 1e3], dtype=torch.float64, device=dev)
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            return float(v.item())

        def gather_only():
            w = sg[0].start()
            if w is not None:
                w.wait()
        gather_only()                                                # (once untimed)
        gather_ms = min(wall(gather_only) for _ in range(reps))
        deint_ms = min(wall(lambda: sg[0].finish()) for _ in range(reps))
        rb_ms = min(wall((lambda: host[0].copy_(surfaces[0].view(-1))) if rank == 0 else (lambda: None)) for _ in range(reps))
        # ---- the gather that was NOT timed, once, against the frame that arrived: hardware evidence for both paths from one run.  Guarded
        #      by a watchdog: a collective that hangs must not cost the line above it ----
        other = {"path": "torch.distributed.gather" if use_native else "sr_rccl_render (grouped ncclSend / ncclRecv, no PyTorch in the data path)"}
        if dist.get_backend() == "nccl" and args.strip_rows == 16:
            import threading

            def other_gather():
                try:
                    torch.cuda.set_device(dev)                        # (a new thread starts on device 0)
                    if use_native:
                        g.render_device(frame, sg[0].local.data_ptr(), stream.cuda_stream)
                        t = time.perf_counter()
                        sg[0].exchange()
                        torch.cuda.synchronize(dev)
                        other["ms"] = (time.perf_counter() - t) * 1e3
                        got = sg[0].full.view(-1) if rank == 0 else None
                    else:
                        native_init()
                        got = torch.zeros(full_px, dtype=torch.int32, device=dev) if rank == 0 else None
                        g.rccl_render(full_frame, got.data_ptr() if rank == 0 else 0, stream.cuda_stream)     # (first call: communicator warm-up)
                        torch.cuda.synchronize(dev)
                        dist.barrier()
                        t = time.perf_counter()
                        g.rccl_render(full_frame, got.data_ptr() if rank == 0 else 0, stream.cuda_stream)
                        torch.cuda.synchronize(dev)
                        other["ms"] = (time.perf_counter() - t) * 1e3
                    if rank == 0:
                        other["frame_equal"] = bool(np.array_equal(got.cpu().numpy(), last_host_frame))
                    other["ok"] = True
                except Exception as e:                                # reported, never fatal for the line
                    other["ok"] = False
                    other["error"] = repr(e)[:300]

            th = threading.Thread(target=other_gather, daemon=True)
            th.start()
            th.join(120.0)
            if th.is_alive():
                other["ok"] = False
                other["error"] = "no completion within 120 s"
                if rank == 0:
                    out["multi_gpu_other_gather"] = other
                    print(json.dumps(out), flush=True)
                os._exit(0 if rank != 0 else 0)
        else:
            other["skipped"] = "needs the nccl backend and 16-row strips"
        if rank == 0:
            out["multi_gpu_other_gather"] = other
            out["multi_gpu"] = {
                "backend": dist.get_backend(), "rccl_ranks": dist.get_world_size() if dist.get_backend() == "nccl" else None,
                "timed_gather": "sr_rccl_render" if use_native else "torch.distributed.gather",
                "world_size": dist.get_world_size(), "visible_devices": ndev,
                "distinct_devices": len({(e["device"]) for e in everyone}),
                "ranks": everyone,
                "render_ms_max": max(e["render_ms"] for e in everyone), "render_ms_min": min(e["render_ms"] for e in everyone),
                "gather_ms": gather_ms, "deinterleave_ms": deint_ms, "readback_ms": rb_ms,
                "bytes_gathered_per_rank": int(sg[0].maxc) * 4,
                "note": "per-rank render_ms = that rank's strips alone (device drained before and after, %d frames); gather / de-interleave / read-back = "
                        "each step alone, max over ranks, best of %d; in the timed region they overlap the next frame's rendering" % (reps, reps),
            }

    if rank == 0 and world == 1 and not args.no_extras:
        # ---- the drop-in call: what the reference's synchronous Renderer.Render() maps to (Renderer.cs:701-778, caller-owned int[] :593):
        #      a BLOCKING sr_render into a plain, pageable, already-touched numpy array; wall time per call ----
        plain = np.zeros(full_px, dtype=np.int32)
        plain[:] = 1
        g.render(frame, out=plain, stats=False)
        torch.cuda.synchronize(dev)
        reps = max(3, min(10, args.steps))
        t1 = time.perf_counter()
        for _ in range(reps):
            g.render(frame, out=plain, stats=False)
        drop_ms = (time.perf_counter() - t1) / reps * 1e3
        out["drop_in_call"] = {"ms": drop_ms, "Mrays_per_s": primary_rays / (drop_ms * 1e-3) / 1e6, "calls": reps,
                               "equals_pipelined_frame": bool(np.array_equal(plain, last_host_frame)),
                               "note": "blocking sr_render(scene, frame, int32* pixels, NULL) into pageable host memory: the frame is enqueued on the library's own "
                                       "stream, every row band is copied back on a copy stream as soon as it is final, the caller's surface is pinned for the call"}

    extras = world == 1 and not in_library and not args.no_extras
    if extras:
        # ---- the same steps with the frame left in HBM (no read-back) ----
        state["readback"] = False
        for _ in range(2):
            step()
        dt = timed(args.steps)
        out["hbm_resident"] = {"value": primary_rays * args.steps / dt / 1e6, "unit": "Mrays/s", "ms_per_step": dt / args.steps * 1e3,
                               "note": "frame stays in HBM (round-1 definition of `value`)"}
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        host[0].copy_(surfaces[0].view(-1))                            # one blocking D2H into pinned memory, for reference
        torch.cuda.synchronize(dev)
        out["d2h_ms"] = (time.perf_counter() - t1) * 1e3

        # ---- per-kernel device time: HIP event pairs around every launch (opt-in), frames rendered as ONE pipeline
        #      (SR_F_NO_SPLIT: same kernels, same work per frame, no two kernels sharing the GPU) ----
        ksteps = min(args.steps, 5)
        frame_ns = sa.Frame.from_buffer_copy(bytes(frame))
        frame_ns.flags |= sa._lib.F_NO_SPLIT
        if frame.area_light_offsets:
            frame_ns.area_light_offsets = frame.area_light_offsets
        local = surfaces[0]
        g.debug_set(sa._lib.DBG_KERNEL_TIMING, 1)
        g.render_device(frame_ns, local.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(dev)
        g.reset_kernel_times()
        t1 = time.perf_counter()
        for _ in range(ksteps):
            g.render_device(frame_ns, local.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(dev)
        ms_unsplit = (time.perf_counter() - t1) / ksteps * 1e3
        kt = {k: (ms / max(1, n), n) for k, (ms, n) in g.kernel_times().items()}     # average ms per launch, launches
        g.debug_set(sa._lib.DBG_KERNEL_TIMING, -1)

        # ---- roofline inputs: deterministic device counters of one (untimed) stats pass ----
        host_px = np.zeros(npix, dtype=np.int32)
        g.render(frame_ns, out=host_px, stats=True)
        rs = g.ray_stats().astype(np.float64)
        out["frame_crc"] = int(np.bitwise_xor.reduce(host_px.view(np.uint32)))
        out["readback_equals_blocking_render"] = bool(np.array_equal(last_host_frame, host_px))
        # algorithmic bytes per LAUNCH (one pipeline = one launch per frame of every kernel) = what the launch must move at
        # least once:
        #   k_primary (packet walk)  4 B pixel per ray + per WAVE 64 B/node + 64 B/camera-cone record + 128 B/FP64 record fetched
        #                            + 64 B queue record per hit
        #   k_shaft (packet walk)    per hit point 64 B queue record + 4 B count; per WAVE 64 B/node + 64 B/TriSlab; 4 B per list entry
        #   k_shadow (k_shadow_cls)  per hit point 64 B + 4 B + 8 B pixel RMW; per candidate 4 B list entry + 64 B TriSlab
        pkt_nodes, pkt_slabs = rs[6] - rs[14], rs[10] - rs[15]
        # the packet walks fetch 128-byte four-wide nodes (64-byte binary nodes with --dbg 13=1, SR_DBG_BVH2_PACKETS)
        S_PKT_NODE = 64 if any(kv.split("=")[0] == "13" and int(kv.split("=")[1]) > 0 for kv in args.dbg) else 128
        bvh_nodes = g.bvh_stats()[1] if args.mode == "bvh" else 0
        algo = {
            "k_primary": S_PIX * npix + rs[2] * S_PKT_NODE + rs[1] * S_SLAB + rs[3] * S_TRI + 64.0 * rs[11],
            "k_shaft": rs[11] * (64.0 + 4.0) + pkt_nodes * S_PKT_NODE + pkt_slabs * S_SLAB + rs[8] * 4.0,
            # one wave per hit point, every lane of a pass fetches its own node / TriSlab: the same records are touched again and again by
            # different hit points, but one launch need not move a record more than once -- capped at the arrays' sizes
            "k_shaft_round2": min(rs[14] * S_NODE + rs[15] * S_SLAB, float(bvh_nodes * S_NODE + args.tris * S_SLAB)),
            "k_shadow": rs[9] * (64.0 + 4.0 + 8.0) + rs[8] * ((S_TRI if args.exact_shadow_tests else S_SLAB) + 4.0),
            # incoherent rays, one per lane: 4 B ray id + 8 B mask RMW per shadow ray, 2 x 64 B queue records + 5 B level colour per mirror
            # ray; every LANE fetches its own 64 B nodes and 128 B FP64 records
            "k_shadow_fallback": rs[16] * 12.0 + rs[18] * S_NODE + rs[17] * S_TRI,
            # (mirror rays walk the four-wide tree privately: a 128-byte node per lane and step)
            "k_bounce": rs[20] * (2 * 64.0 + 5.0) + rs[22] * S_PKT_NODE + rs[21] * S_TRI,
        }
        dom = max(kt.items(), key=lambda kv: kv[1][0]) if kt else ("none", (float("nan"), 0))
        dom_name, dom_ms = dom[0], dom[1][0]
        algo_bytes = algo.get(dom_name, float("nan"))
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms == dom_ms and dom_ms > 0 else float("nan")
        prof = committed_profile(profile_dir(args))
        tkey = "%s_%d_%d_%d" % (args.mode, args.tris, args.res, args.shadows) + ("_b%d" % args.bounces if args.bounces > 0 else "")
        traffic = (prof["traffic"].get(tkey) or {}).get(dom_name)
        prof_kernels = (prof["issue"] or {}).get("kernels") or {}
        prof_name = {"k_shaft": "k_shaft_pkt", "k_shadow": "k_shadow_cls", "k_shadow_fallback": "k_shadow_rays", "k_shaft_round2": "k_shaft"}.get(dom_name, dom_name)
        if dom_name == "k_bounce" and "k_bounce_walk" in prof_kernels:
            prof_name = "k_bounce_walk"                                  # the level's walk kernel (prepare / finish run at full lanes)
        issue_achieved = prof_kernels.get(prof_name, {}).get("useful_lane_ops_per_s")
        # ---- measured copy bandwidth of THIS device ("use the measured number as the denominator and state both"): a 1 GiB
        #      device-to-device copy of float4 elements, best of 5 ----
        peak_measured = None
        try:
            src = torch.empty(1 << 26, 4, dtype=torch.float32, device=dev).normal_()
            dst = torch.empty_like(src)
            dst.copy_(src)
            best = float("inf")
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                dst.copy_(src)
                b.record(stream)
                b.synchronize()
                best = min(best, a.elapsed_time(b))
            peak_measured = 2.0 * src.numel() * 4 / (best * 1e-3) / 1e9
            del src, dst
        except Exception:
            pass
        # compulsory traffic: every scene byte once + every pixel once, per primary ray
        # FP64 records (leaf order), TriSlab + CamCone records, binary nodes + the camera- and the light-ordered four-wide copies (~ half as many nodes, twice the size each)
        scene_bytes = args.tris * (S_TRI + 2 * S_SLAB) + bvh_nodes * S_NODE * 3
        compulsory = (scene_bytes + 4.0 * npix) / primary_rays
        out.update({
            "rays": {"primary": rs[0], "shadow": rs[4], "primary_plus_shadow_Mrays_per_s": (rs[0] + rs[4]) / (ms_per_step * 1e-3) / 1e6},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if achieved == achieved else None, "traffic": traffic,
                         "peak_measured": peak_measured, "frac_of_peak_measured": (achieved / peak_measured) if (peak_measured and achieved == achieved) else None,
                         "served_by_cache": bool(achieved == achieved and achieved > (peak_measured or HBM_PEAK_GBS)),
                         "compulsory_bytes_per_ray": compulsory, "scene_bytes": scene_bytes,
                         "compulsory_GBs_at_this_rate": compulsory * primary_rays / (ms_per_step * 1e-3) / 1e9,
                         "kernel": dom_name, "kernel_ms_per_launch": dom_ms, "launches_timed": dom[1][1], "algorithmic_bytes_per_launch": algo_bytes,
                         "all_kernels_ms_per_launch": {k: v[0] for k, v in kt.items()},
                         "all_kernels_algorithmic_GBs": {k: algo[k] / (v[0] * 1e-3) / 1e9 for k, v in kt.items() if k in algo and v[0] > 0},
                         "note": "achieved = algorithmic bytes of ONE launch of the dominant kernel / its average launch duration (HIP event pairs on the "
                                 "launch stream, frames rendered as one pipeline so that no two kernels share the GPU).  The scene (128 MB records + BVH + "
                                 "2 x 64 MB fp32 records at 1 M triangles) is largely cache resident: `traffic` (HBM bytes from the FETCH_SIZE / WRITE_SIZE counters of "
                                 "the committed rocprofv3 passes) is what actually crossed the HBM interface; `served_by_cache` marks a kernel whose algorithmic "
                                 "bytes per second exceed the peak (per-lane gathers of records other lanes fetched a moment ago).  The kernels are bound by "
                                 "instruction issue or gather latency (see roofline_issue), not by HBM"},
            "roofline_issue": {"bound": "valu_issue", "achieved": issue_achieved, "peak": VALU_PEAK_LANE_OPS, "unit": "lane-instructions/s",
                               "frac": (issue_achieved / VALU_PEAK_LANE_OPS) if issue_achieved else None, "kernel": prof_name,
                               "peak_derivation": "256 CUs x 4 SIMDs x 16 lanes/cycle x 2.4 GHz (an FMA counts once; packed fp32 counts once per lane)",
                               "kernels": prof_kernels or None, "source": (prof["issue"] or {}).get("source"),
                               "note": "achieved = SQ_THREAD_CYCLES_VALU (active lanes summed over VALU instructions) / kernel time of the dominant kernel, from the "
                                       "committed rocprofv3 passes of this workload"},
            "ms_per_step_one_pipeline": ms_unsplit,
            "pipeline_counters": g.debug_counters(), "device_counters": [float(x) for x in rs],
            "shadow_pairs": {"classified_fp32": rs[12], "decided_fp64": rs[13], "fp64_fraction": (rs[13] / rs[12]) if rs[12] else None},
        })

        # ---- the same frame without ShadowMethod (rayTraceShadows = false): the "primary rays only" rate ----
        if args.shadows > 0:
            a0 = copy.copy(args)
            a0.shadows = 0
            f0 = make_frame(a0)
            for _ in range(2):
                g.render_device(f0, local.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                g.render_device(f0, local.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize(dev)
            dt0 = (time.perf_counter() - t1) / args.steps
            out["primary_only"] = {"value": primary_rays / dt0 / 1e6, "unit": "Mrays/s", "ms_per_step": dt0 * 1e3,
                                   "note": "same scene, pose and resolution with rayTraceShadows = false (shading on), frame left in HBM"}
        # ---- the two surface passes of Renderer.Render() (PostProcessImage / AntiAliasImage) on the resident frame ----
        n = args.res * args.res
        aa_dst = torch.empty(n // 4, dtype=torch.int32, device=dev)
        g.debug_set(sa._lib.DBG_KERNEL_TIMING, 1)
        g.reset_kernel_times()
        reps = 20
        for _ in range(reps):
            g.post_process_device(local.data_ptr(), n, 1, 0, stream.cuda_stream)                    # Style.ColorShuffle, in place
            g.anti_alias_device(local.data_ptr(), args.res // 2, args.res // 2, 2, aa_dst.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(dev)
        kp = g.kernel_times()
        g.debug_set(sa._lib.DBG_KERNEL_TIMING, -1)
        pp_ms = kp["k_post_process"][0] / reps
        aa_ms = kp["k_anti_alias"][0] / reps
        out["surface_passes"] = {
            "post_process": {"ms": pp_ms, "achieved_GBps": 8.0 * n / (pp_ms * 1e-3) / 1e9, "bytes": "8 B/pixel (read + write in place)"},
            "anti_alias_2x": {"ms": aa_ms, "achieved_GBps": 5.0 * n / (aa_ms * 1e-3) / 1e9, "bytes": "4 B/source pixel + 4 B/destination pixel"},
            "peak_GBps": 8000.0,
        }
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, v9, argb, bmin, bmax)
    if (args.verify or (n_gpus > 1 and not args.no_verify)) and rank == 0:
        ref_scene = g
        if in_library:
            ref_scene = sa.GpuScene(0)
            ref_scene.set_triangles(v9, argb, bmin, bmax)
            ref_scene.build((sa.MODE_BVH,))
        ref = np.zeros(args.res * args.res, dtype=np.int32)
        ref_scene.render(make_frame(args), out=ref, stats=False)
        out["verify"] = {"full_frame_equal": bool(np.array_equal(last_host_frame, ref)), "crc": int(np.bitwise_xor.reduce(ref.view(np.uint32)))}
        assert out["verify"]["full_frame_equal"], "the frame that arrived on the host differs from the single-GPU frame"
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
