"""Kernel timeline of ONE frame of rank 0's share of the N-way strip split (run under rocprofv3 --kernel-trace):
   python scripts/gpu_timeline.py render N [key=value ...]   renders 6 frames, marks nothing (the trace's last frame is read)
   python scripts/gpu_timeline.py read <kernel_trace.csv>    prints start offset / duration of every kernel of the last frame"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "render":
    import numpy as np, torch
    import softray_amd as sa
    import bench
    bench._imports()
    n = int(sys.argv[2])
    args = bench.argparse.Namespace(res=4096, tris=1000000, shadows=100, spp=1, mode="bvh", depth=1.5, extent=0.05, bounces=0,
                                    reflectivity=0.0, strip_rows=16, static_shadows=False)
    v9, argb = sa.make_random_triangles(args.tris, 12345, space=1.0 - args.extent, extent=args.extent, origin=-0.5, opaque=True)
    g = sa.GpuScene(0)
    g.set_triangles(v9, argb, np.array([-0.5] * 3), np.array([0.5] * 3))
    g.build((sa.MODE_BVH,))
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        g.debug_set(int(k), int(v))
    f = bench.make_frame(args, (16, n, 0) if n > 1 else None)
    buf = torch.empty(g.pixel_count(f), dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream()
    drain = os.environ.get("DRAIN", "1") == "1"              # DRAIN=0: frames back to back, as the bench issues them
    for i in range(6):
        g.render_device(f, buf.data_ptr(), s.cuda_stream)
        if drain: torch.cuda.synchronize()
    torch.cuda.synchronize()
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # frames are separated by a device drain: the last gap > 200 us splits off the last frame
    starts = [int(r["Start_Timestamp"]) for r in rows]
    ends = [int(r["End_Timestamp"]) for r in rows]
    cut = 0
    prim = [i for i, r in enumerate(rows) if "k_primary" in r["Kernel_Name"]]
    if len(prim) >= 6: cut = prim[-6]                    # the last three frames (two k_primary each)
    t0 = starts[cut]
    busy_end = t0
    for r in rows[cut:]:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = a - busy_end
        busy_end = max(busy_end, b)
        print("%8.1f us  +%7.1f us  idle-before %6.1f  %s" % ((a - t0) / 1e3, (b - a) / 1e3, max(gap, 0) / 1e3, r["Kernel_Name"][:70]))
    print("frame %.1f us" % ((busy_end - t0) / 1e3))
