"""Later shaft rounds: one wave per hit point (k_shaft_coop) against private per-lane walks (k_shaft), headline scene.
usage: python scripts/gpu_round2.py [res ...]   -- prints kernel times (one pipeline), counters and whether the frames agree"""
import json, os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import softray_amd as sa
import bench
bench._imports()
v9, argb = sa.make_random_triangles(1000000, 12345, space=0.95, extent=0.05, origin=-0.5, opaque=True)
g = sa.GpuScene(0)
g.set_triangles(v9, argb, np.array([-0.5] * 3), np.array([0.5] * 3))
g.build((sa.MODE_BVH,))
for res in [int(a) for a in sys.argv[1:]] or [1024, 4096]:
    args = bench.argparse.Namespace(res=res, tris=1000000, shadows=100, spp=1, mode="bvh", depth=1.5, extent=0.05, bounces=0,
                                    reflectivity=0.0, strip_rows=16, static_shadows=False)
    f = bench.make_frame(args, None)
    f.flags |= sa._lib.F_NO_SPLIT
    buf = torch.empty(g.pixel_count(f), dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream()
    out = {}
    for name, v in (("private", 2), ("coop", 0)):
        g.debug_set(sa._lib.DBG_PER_LANE_SHAFT, v)
        g.debug_set(sa._lib.DBG_KERNEL_TIMING, 1)
        for _ in range(2):
            g.reset_kernel_times()
            g.render_device(f, buf.data_ptr(), s.cuda_stream)
            torch.cuda.synchronize()
        kt = {k: round(t[0], 3) for k, t in g.kernel_times().items()}
        g.debug_set(sa._lib.DBG_KERNEL_TIMING, -1)
        out[name] = {"kernels_ms": kt, "counters": [int(c) for c in g.debug_counters()], "crc": zlib.crc32(buf.cpu().numpy().tobytes())}
        print(json.dumps({"res": res, name: out[name]}), flush=True)
    print(json.dumps({"res": res, "frames_equal": len({o["crc"] for o in out.values()}) == 1}), flush=True)
