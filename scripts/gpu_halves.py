import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import softray_amd as sa, bench
args = bench.argparse.Namespace(res=4096, tris=1000000, shadows=100, spp=1, mode="bvh", depth=1.5, extent=0.05, bounces=0, reflectivity=0.0, strip_rows=16, static_shadows=False)
v9, argb = sa.make_random_triangles(args.tris, 12345, space=0.95, extent=0.05, origin=-0.5, opaque=True)
g = sa.GpuScene(0); g.set_triangles(v9, argb, np.array([-0.5]*3), np.array([0.5]*3)); g.build((sa.MODE_BVH,))
buf = torch.empty(4096*4096, dtype=torch.int32, device='cuda'); s = torch.cuda.current_stream()
for a, b in ((0, 4095), (0, 2047), (2048, 4095), (0, 1023), (1024, 2047), (2048, 3071), (3072, 4095)):
    f = bench.make_frame(args); f.start_row, f.end_row = a, b; f.flags |= sa._lib.F_NO_SPLIT
    for _ in range(2): g.render_device(f, buf.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): g.render_device(f, buf.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize(); print(a, b, round((time.perf_counter()-t)/5*1e3, 2), 'ms')
