#!/usr/bin/env python3
"""bench.py -- headline benchmark of the raytrace hot path on MI355X.

Metric (BASELINE.json): Mrays/s = PRIMARY rays per second = W*H*subPixelRes^2 / t_frame, on the
1M-random-triangle scene (System.Random seed 12345) at 4096x4096, shading + the
reference's 100-sample dynamic soft shadows ("primary+shadow"), traced through the library's BVH.
A "step" is one full frame.  The scene, the BVH and the frame constants are resident in HBM before
the timed region; the frame stays in HBM (the PCIe-inclusive rate is reported separately).

    python bench.py --gpus N --steps K --warmup W

N > 1 (launched by torch.distributed.run, one rank per GPU): the frame is row-tiled in interleaved
16-row strips, every rank renders its strips into a compact device buffer and ONE RCCL gather over
xGMI collects the strips on rank 0 ("scaling": "strong" -- the frame is fixed).
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

import softray_amd as sa
from softray_amd import renderer as R
from softray_amd.distributed import StripGather

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
S_NODE, S_TRI, S_PIX = 64, 128, 4   # bytes: BVH node (two fp32 child boxes + links), triangle record, pixel store


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


def cpu_baseline(args, v9, argb, bmin, bmax, budget_s=20.0):
    """The reference's CPU path = the C++ oracle (statement-level restatement; real C# cannot be built here),
    reference tree depth 15 / 25 per leaf, row-block threads on all host cores, timed on a centred band of
    rows of the SAME frame (bounded sample)."""
    from oracle import oracle_py as orc                              # the only place bench.py touches oracle/
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                            # honour a cgroup CPU quota (the GPU box gives each job a share)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(math.ceil(int(q) / int(per)))))
    except Exception:
        pass
    o = orc.Scene()
    o.set_triangles(v9, argb, bmin, bmax)
    t0 = time.time()
    assert o.build_tree() == 0
    build_s = time.time() - t0
    f = orc.Frame.from_buffer_copy(bytes(make_frame(args)))
    f.trace_mode = orc.MODE_REF_TREE
    mid = args.res // 2
    rows, total_rows, total_s = 1, 0, 0.0
    scratch = np.zeros(args.res * args.res, dtype=np.int32)
    while True:                                                      # grow the band until ~budget_s of CPU work
        f.start_row, f.end_row = mid - rows // 2, mid - rows // 2 + rows - 1
        t0 = time.time()
        o.render(f, threads=cores, out=scratch)
        dt = time.time() - t0
        total_rows, total_s = rows, dt
        if dt >= budget_s * 0.5 or rows >= args.res:
            break
        rows = min(args.res, max(rows * 2, int(rows * budget_s / max(dt, 1e-3) * 0.8)))
    rays = total_rows * args.res * args.spp * args.spp
    return {"value": rays / total_s / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d centred rows x %d cols of the same %dx%d frame (%d primary rays, %.1f s); C++ restatement of "
                      "the reference algorithm (reference tree 15/25, row-block threads) -- real C# unavailable; "
                      "tree build %.1f s excluded" % (total_rows, args.res, args.res, args.res, rays, total_s, build_s)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
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
    ap.add_argument("--device-build", action="store_true", help="build the BVH on the GPU (LBVH) instead of the host SAH builder")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact-shadow-tests", action="store_true", help="k_shadow_test (every pair in FP64) instead of the fp32-classified k_shadow_cls")
    ap.add_argument("--dbg", action="append", default=[], metavar="KEY=VALUE", help="sr_debug_set hook, e.g. --dbg 1=16 (SR_DBG_ROUND_CAP0 = 16)")
    ap.add_argument("--verify", action="store_true", help="after timing: rank 0 re-renders the whole frame alone and compares it with the gathered one")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU rehearsals)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU implementation")
    local_rank = local_rank % ndev            # (rehearsals with more ranks than GPUs share devices; the driver uses one rank per GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    # ---- scene resident in HBM (replicated on every rank: 128 MB of records) ----
    v9, argb = sa.make_random_triangles(args.tris, 12345, space=1.0 - args.extent, extent=args.extent, origin=-0.5, opaque=True)
    bmin, bmax = np.array([-0.5] * 3), np.array([0.5] * 3)
    g = sa.GpuScene(local_rank)
    g.set_triangles(v9, argb, bmin, bmax)
    if args.exact_shadow_tests:
        g.debug_set(sa._lib.DBG_EXACT_SHADOW_TESTS, 1)
    for kv in args.dbg:
        k, v = kv.split("=")
        g.debug_set(int(k), int(v))
    t0 = time.time()
    g.build(({"bvh": sa.MODE_BVH, "ref": sa.MODE_REF_TREE}.get(args.mode),) if args.mode != "brute" else (),
            on_device=args.device_build and args.mode == "bvh")
    build_s = time.time() - t0

    strips = (args.strip_rows, world, rank) if world > 1 else None
    frame = make_frame(args, strips)
    npix = g.pixel_count(frame)
    stream = torch.cuda.current_stream(dev)
    sg = StripGather(args.res, args.res, args.strip_rows, world, rank, dev) if world > 1 else None
    local = sg.local if sg else torch.empty(npix, dtype=torch.int32, device=dev)
    if sg:
        assert sg.counts[rank] == npix

    def step():
        if args.static_shadows:
            g.reset_shadow_cache()                                   # every step generates the whole cache again
        g.render_device(frame, local.data_ptr(), stream.cuda_stream)
        if sg:
            sg.exchange()                                             # RCCL gather over xGMI + de-interleave on rank 0

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    g.reset_kernel_times()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # per-kernel device time: one HIP event pair per launch on the launch stream, recorded inside the library; average =
    # total / launches.  The timed steps above run the frame as two concurrent halves (two internal streams), so their
    # kernels share the GPU pairwise; the durations the roofline uses come from K more frames rendered as ONE pipeline
    # (SR_F_NO_SPLIT: same kernels, same work per frame, no overlap), timed as a whole for reference
    frame_ns = sa.Frame.from_buffer_copy(bytes(frame))
    frame_ns.flags |= sa._lib.F_NO_SPLIT
    if frame.area_light_offsets:
        frame_ns.area_light_offsets = frame.area_light_offsets
    g.debug_set(sa._lib.DBG_KERNEL_TIMING, 1)                         # event pairs around every launch: only in this extra pass
    g.render_device(frame_ns, local.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)
    g.reset_kernel_times()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        g.render_device(frame_ns, local.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(dev)
    ms_unsplit = (time.perf_counter() - t1) / args.steps * 1e3
    kt = {k: (ms / max(1, n), n) for k, (ms, n) in g.kernel_times().items()}
    g.debug_set(sa._lib.DBG_KERNEL_TIMING, -1)

    primary_rays = args.res * args.res * args.spp * args.spp
    value = primary_rays * args.steps / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    out = None
    if rank == 0:
        # ---- roofline inputs: deterministic device counters of one (untimed) stats pass over this rank's share ----
        host_px = np.zeros(npix, dtype=np.int32)
        g.render(frame, out=host_px, stats=True)
        rs = g.ray_stats().astype(np.float64)
        # algorithmic bytes per FRAME of each kernel, at wave granularity (what a launch must move at least once):
        #   k_primary      per ray (lanes walk independently): 4 B pixel + 64 B/node visit + 128 B/triangle test (+64 B queue record per hit)
        #   k_shaft        per hit point: 64 B queue record + 64 B/node visit + 64 B/slab record + 4 B/list entry written
        #   k_shadow(_test) per hit point: 64 B queue record + 4 B count + per staged triangle (4 B list entry + 128 B record) + 8 B pixel RMW
        per_ray_touch = S_PIX * (npix * 1.0) + (rs[2] + rs[6]) * S_NODE + (rs[1] + rs[5]) * S_TRI     # SURVEY 8d figure (per-lane touches)
        algo = {
            "k_primary": S_PIX * npix + rs[2] * S_NODE + rs[1] * S_TRI + 64.0 * (rs[4] / max(1, args.shadows) if args.shadows else 0),
            "k_shaft": rs[11] * 64.0 + rs[6] * S_NODE + rs[10] * 64.0 + rs[8] * 4.0,
            # k_shadow_cls reads the 64-byte fp32 TriSlab of a candidate (k_shadow_test, the FP64 cross-check, its 128-byte record)
            "k_shadow": rs[9] * (64.0 + 4.0 + 8.0) + rs[8] * ((S_TRI if args.exact_shadow_tests else 64.0) + 4.0),
            "k_render": per_ray_touch,
        }
        tot = {k: v[0] * v[1] for k, v in kt.items()}                 # total ms per kernel over the timed steps
        fam = lambda k: "k_shaft" if k.startswith("k_shaft") else ("k_shadow" if k.startswith("k_shadow") else k)
        fam_ms = {}
        for k, v in tot.items():
            fam_ms[fam(k)] = fam_ms.get(fam(k), 0.0) + v / args.steps  # ms per frame
        dom = max(fam_ms.items(), key=lambda kv: kv[1]) if fam_ms else ("none", float("nan"))
        dom_ms = dom[1]
        algo_bytes = algo.get(dom[0], per_ray_touch)
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms == dom_ms and dom_ms > 0 else float("nan")
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("%s_%d_%d_%d" % (args.mode, args.tris, args.res, args.shadows), {}).get(dom[0])
            except Exception:
                traffic = None
        # what actually bounds the kernels: SQ counters of the committed profile of this workload (separate rocprofv3 --pmc passes)
        issue = None
        spath = os.path.join(ROOT, "profiles", "r01d_pipeline", "sq_counters_summary.csv")
        if os.path.exists(spath) and traffic is not None:
            try:
                import csv
                issue = {"source": "profiles/r01d_pipeline/sq_counters_summary.csv (rocprofv3 --pmc SQ_*, same workload)",
                         "valu_busy_frac": {}, "active_lanes_per_valu_inst": {}, "wait_any_frac_of_wave_cycles": {}}
                for r in csv.DictReader(open(spath)):
                    if r["kernel"] in ("k_primary", "k_shaft", "k_shadow_test"):
                        issue["valu_busy_frac"][r["kernel"]] = float(r["valu_busy_frac_at_2.4GHz_1024_SIMDs"])
                        issue["active_lanes_per_valu_inst"][r["kernel"]] = float(r["active_lanes_per_valu_inst"])
                        issue["wait_any_frac_of_wave_cycles"][r["kernel"]] = float(r["SQ_WAIT_ANY"]) / float(r["SQ_WAVE_CYCLES"])
            except Exception:
                issue = None
        # PCIe-inclusive frame time (never `value`): one D2H of the frame
        t1 = time.perf_counter()
        local.cpu()
        d2h_ms = (time.perf_counter() - t1) * 1e3
        out = {
            "metric": "Mrays/s (primary rays) at %dx%d, shading + %s, %s" % (
                args.res, args.res,
                "100-sample soft shadows" if args.shadows == 100 else ("no shadows" if args.shadows == 0 else "%d-sample shadows" % args.shadows),
                {"bvh": "own BVH", "ref": "reference tree", "brute": "brute force"}[args.mode]),
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d random triangles (System.Random seed 12345, extent %g, unit cube) + BVH, %dx%d, "
                                   "spp %d, shading + %d shadow samples/hit, pose yaw135/pitch-22/depth %.1f" % (
                                       args.tris, args.extent, args.res, args.res, args.spp * args.spp, args.shadows, args.depth),
                       "trace_mode": args.mode, "parallelism": "rows x%d (interleaved %d-row strips)" % (world, args.strip_rows)},
            "rays_rank0": {"primary": rs[0], "shadow": rs[4], "tri_tests": rs[1] + rs[5], "node_visits": rs[2] + rs[6],
                           "primary_plus_shadow_Mrays_per_s": ((rs[0] + rs[4]) / (ms_per_step * 1e-3) / 1e6) if world == 1 else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if achieved == achieved else None, "traffic": traffic,
                         "kernel": dom[0], "kernel_ms_per_frame": dom_ms, "algorithmic_bytes_per_frame": algo_bytes,
                         "launches_per_frame": sum(v[1] for k, v in kt.items() if fam(k) == dom[0]) / args.steps,
                         "all_kernels_ms_per_frame": fam_ms,
                         "all_kernels_algorithmic_GBs": {k: algo.get(k, 0.0) / (v * 1e-3) / 1e9 for k, v in fam_ms.items() if v > 0},
                         "per_ray_touch_bytes_per_frame": per_ray_touch, "issue_bound": issue,
                         "note": "achieved = wave-granular algorithmic bytes of the dominant kernel family per frame / its device time per frame "
                                 "(HIP event pairs around every launch of K frames rendered as one pipeline, SR_F_NO_SPLIT, so that no two kernels share "
                                 "the GPU; `value` is measured with the default two concurrent half-frame pipelines).  The scene (128 MB records + 21 MB BVH + 64 MB slabs) is cache "
                                 "resident and the kernels are FP64-issue / latency bound, not HBM bound; per_ray_touch is SURVEY 8d's "
                                 "per-lane figure (4 B + 64 B/node + 128 B/triangle test per ray)"},
            "kernels_ms": {k: v[0] for k, v in kt.items()}, "kernel_launches": {k: v[1] for k, v in kt.items()},
            "ms_per_step_one_pipeline": ms_unsplit,
            "pipeline_counters_last_band": g.debug_counters(), "device_counters": [float(x) for x in rs],
            "shadow_pairs": {"classified_fp32": rs[12], "decided_fp64": rs[13], "fp64_fraction": (rs[13] / rs[12]) if rs[12] else None},
            "build_s": build_s, "d2h_ms": d2h_ms,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, v9, argb, bmin, bmax)
    if args.verify and rank == 0:
        ref = np.zeros(args.res * args.res, dtype=np.int32)
        g.render(make_frame(args), out=ref, stats=False)
        got = (sg.full if sg else local).cpu().numpy().reshape(-1)
        out["verify"] = {"full_frame_equal": bool(np.array_equal(got, ref)), "crc": int(np.bitwise_xor.reduce(ref.view(np.uint32)))}
        assert out["verify"]["full_frame_equal"], "gathered frame differs from the single-process frame"
    # ---- the same frame without ShadowMethod (rayTraceShadows = false): the "primary rays only" rate, same scene/pose ----
    if world == 1 and args.shadows > 0:
        import copy
        a0 = copy.copy(args)
        a0.shadows = 0
        f0 = make_frame(a0)
        for _ in range(max(1, args.warmup)):
            g.render_device(f0, local.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            g.render_device(f0, local.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(dev)
        dt0 = (time.perf_counter() - t1) / args.steps
        out["primary_only"] = {"value": primary_rays / dt0 / 1e6, "unit": "Mrays/s", "ms_per_step": dt0 * 1e3,
                               "note": "same scene, pose and resolution with rayTraceShadows = false (shading on): primary rays only"}
    # ---- the two surface passes of Renderer.Render() (PostProcessImage / AntiAliasImage) on the resident frame ----
    if world == 1 and rank == 0:
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
        pp_ms = kp["k_post_process"][0] / reps
        aa_ms = kp["k_anti_alias"][0] / reps
        out["surface_passes"] = {
            "post_process": {"ms": pp_ms, "achieved_GBps": 8.0 * n / (pp_ms * 1e-3) / 1e9, "bytes": "8 B/pixel (read + write in place)"},
            "anti_alias_2x": {"ms": aa_ms, "achieved_GBps": 5.0 * n / (aa_ms * 1e-3) / 1e9, "bytes": "4 B/source pixel + 4 B/destination pixel"},
            "peak_GBps": 8000.0,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
