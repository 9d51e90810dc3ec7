"""Host SAH build (all host cores) vs the device LBVH build: build time, identical pixels, frame time of the headline pose.
usage: python scripts/gpu_lbvh.py [triangles] [extent]"""
import json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import softray_amd as sa
import bench
bench._imports()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
extent = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
args = bench.argparse.Namespace(res=4096, tris=n, shadows=100, spp=1, mode="bvh", depth=1.5, extent=extent, bounces=0,
                                reflectivity=0.0, strip_rows=16, static_shadows=False)
v9, argb = sa.make_random_triangles(n, 12345, space=1.0 - extent, extent=extent, origin=-0.5, opaque=True)
g = sa.GpuScene(0)
g.set_triangles(v9, argb, np.array([-0.5] * 3), np.array([0.5] * 3))
f = bench.make_frame(args, None)
buf = torch.empty(g.pixel_count(f), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream()
out = {}
for name, dev in (("host_sah", False), ("device_lbvh", True)):
    t = time.perf_counter()
    g.build((sa.MODE_BVH,), on_device=dev)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t
    if dev:                                              # second build: buffers are allocated, code objects loaded
        t = time.perf_counter()
        g.build((sa.MODE_BVH,), on_device=True)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t
    for _ in range(2):
        g.render_device(f, buf.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        g.render_device(f, buf.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 5 * 1e3
    out[name] = {"build_s": round(build_s, 3), "frame_ms": round(ms, 2), "bvh_stats": [int(x) for x in g.bvh_stats()],
                 "counters": [int(c) for c in g.debug_counters()], "crc": zlib.crc32(buf.cpu().numpy().tobytes())}
    print(json.dumps({name: out[name]}), flush=True)
print(json.dumps({"triangles": n, "frames_equal": len({o["crc"] for o in out.values()}) == 1}))
