"""How many of the shaft walk's node steps happen in the top three / four levels of the four-wide tree (level-ordered nodes < 21 / < 85):
the upper bound of what sharing the upper tree between walks could save.  Needs the STATS instantiation (one counted frame)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, bench
bench._imports()
import softray_amd as sa
args = bench.argparse.Namespace(res=4096, tris=1000000, shadows=100, spp=1, mode="bvh", depth=1.5, extent=0.05, bounces=0, reflectivity=0.0, strip_rows=16, static_shadows=False, no_split=True)
v9, argb = sa.make_random_triangles(args.tris, 12345, space=0.95, extent=0.05, origin=-0.5, opaque=True)
g = sa.GpuScene(0); g.set_triangles(v9, argb, np.array([-0.5]*3), np.array([0.5]*3)); g.build((sa.MODE_BVH,))
f = bench.make_frame(args)
out = np.zeros(4096*4096, dtype=np.int32)
g.render(f, out=out, stats=True)
rs = g.ray_stats(); c = g.debug_counters()
print("node steps", int(rs[6]-rs[14]), "top3 levels (<21)", c[4], "top4 levels (<85)", c[5], "hit points", int(rs[11]), "depth", g.bvh_stats())
