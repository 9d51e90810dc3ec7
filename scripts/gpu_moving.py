"""Cost of the per-(camera, light) pre-passes: the headline frame with a camera (and optionally a light) that moves EVERY frame
(k_facing_partition + k_cam_cones + 2 x k_order_nodes run per frame) against the static camera.  Frames stay in HBM."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import softray_amd as sa
import bench
bench._imports()
args = bench.argparse.Namespace(res=4096, tris=int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, shadows=100, spp=1, mode="bvh", depth=1.5,
                                extent=float(sys.argv[2]) if len(sys.argv) > 2 else 0.05, bounces=0, reflectivity=0.0, strip_rows=16, static_shadows=False)
v9, argb = sa.make_random_triangles(args.tris, 12345, space=1.0 - args.extent, extent=args.extent, origin=-0.5, opaque=True)
g = sa.GpuScene(0)
g.set_triangles(v9, argb, np.array([-0.5] * 3), np.array([0.5] * 3))
g.build((sa.MODE_BVH,))
buf = torch.empty(args.res * args.res, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream()

def frame(yaw_deg, light_shift=0.0):
    f = bench.make_frame(args)
    t, it = sa.instance_matrices([0.0, 0.0, args.depth], yaw_deg / 180.0 * math.pi, -22.0 / 180.0 * math.pi, 0.0)
    for i in range(12):
        f.transform[i] = t[i]
        f.inv_transform[i] = it[i]
    f.light_pos_view[0] += light_shift
    return f

def run(frames):
    for f in frames[:3]:
        g.render_device(f, buf.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in frames:
        g.render_device(f, buf.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / len(frames) * 1e3

n = 24
out = {"triangles": args.tris,
       "static_ms": run([frame(135.0)] * n),
       "camera_moves_every_frame_ms": run([frame(135.0 + 0.25 * k) for k in range(n)]),
       "camera_and_light_move_every_frame_ms": run([frame(135.0 + 0.25 * k, 0.01 * k) for k in range(n)])}
print(json.dumps(out))
