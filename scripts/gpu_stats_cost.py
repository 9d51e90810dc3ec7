"""What do the statistics of a reference-tree frame cost?  obj.3DS 1024^2 + 100-sample shadows: no statistics / primary only / all."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import softray_amd as sa
from helpers import make_frame, GOLDEN

g = sa.GpuScene(0)
g.load_3ds(open(os.path.join(GOLDEN, "obj.3ds"), "rb").read())
g.build((sa.MODE_REF_TREE, sa.MODE_BVH))
out = np.zeros(1024 * 1024, dtype=np.int32)
for shadows in (False, True):
    for name, mode, stats, flag in (("bvh, no stats", sa.MODE_BVH, False, 0), ("ref tree, no stats", sa.MODE_REF_TREE, False, 0),
                                    ("ref tree, primary stats only", sa.MODE_REF_TREE, True, sa._lib.F_PRIMARY_STATS_ONLY),
                                    ("ref tree, all stats", sa.MODE_REF_TREE, True, 0)):
        f = sa.Frame.from_buffer_copy(bytes(make_frame(1024, shadows=shadows)))
        f.trace_mode = mode
        f.flags |= flag
        g.debug_set(sa._lib.DBG_KERNEL_TIMING, 0)
        for _ in range(3):
            g.render(f, out=out, stats=stats)
        t = time.perf_counter()
        for _ in range(10):
            g.render(f, out=out, stats=stats)
        dt = (time.perf_counter() - t) / 10
        g.debug_set(sa._lib.DBG_KERNEL_TIMING, 1)
        g.reset_kernel_times()
        nosplit = sa.Frame.from_buffer_copy(bytes(f)); nosplit.flags |= sa._lib.F_NO_SPLIT
        g.render(nosplit, out=out, stats=stats)
        print("shadows=%d %-30s %.3f ms  kernels %s" % (shadows, name, dt * 1e3, {k: round(v[0], 3) for k, v in g.kernel_times().items()}), flush=True)
