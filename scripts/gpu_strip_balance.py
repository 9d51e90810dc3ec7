"""Per-rank cost of the N-way strip split of the headline frame, measured on one GPU: renders each rank's strips alone."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import softray_amd as sa
import bench
bench._imports()
args = bench.argparse.Namespace(res=4096, tris=1000000, shadows=100, spp=1, mode="bvh", depth=1.5, extent=0.05, bounces=0,
                                reflectivity=0.0, strip_rows=16, static_shadows=False)
v9, argb = sa.make_random_triangles(args.tris, 12345, space=1.0 - args.extent, extent=args.extent, origin=-0.5, opaque=True)
g = sa.GpuScene(0)
g.set_triangles(v9, argb, np.array([-0.5] * 3), np.array([0.5] * 3))
g.build((sa.MODE_BVH,))
for kv in sys.argv[1:]:                                  # sr_debug_set hooks: key=value
    k, v = kv.split("=")
    g.debug_set(int(k), int(v))
out = {}
STRIP = int(os.environ.get("STRIP_ROWS", "16"))
for n in (1, 8):
    times = []
    for k in range(n):
        f = bench.make_frame(args, (STRIP, n, k) if n > 1 else None)
        buf = torch.empty(g.pixel_count(f), dtype=torch.int32, device="cuda")
        s = torch.cuda.current_stream()
        for _ in range(2):
            g.render_device(f, buf.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(6):
            g.render_device(f, buf.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t) / 6 * 1e3)
        if k == 0:                                       # kernel times of one rank's share, as ONE pipeline (no overlapping kernels)
            g.debug_set(sa._lib.DBG_KERNEL_TIMING, 1)
            g.reset_kernel_times()
            f.flags |= sa._lib.F_NO_SPLIT
            g.render_device(f, buf.data_ptr(), s.cuda_stream)
            torch.cuda.synchronize()
            kt = {kk: round(v[0], 2) for kk, v in g.kernel_times().items()}
            g.debug_set(sa._lib.DBG_KERNEL_TIMING, -1)
    out[n] = {"max_ms": max(times), "min_ms": min(times), "speedup_bound": out[1]["max_ms"] / max(times) if n > 1 else 1.0, "kernels_rank0_ms": kt,
              "rank_ms": [round(x, 3) for x in times]}
print(json.dumps(out))
