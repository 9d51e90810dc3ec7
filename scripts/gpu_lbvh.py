import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import softray_amd as sa
v9, argb, bmin, bmax = sa.unit_cube_scene(1000000)
g = sa.GpuScene(0); g.set_triangles(v9, argb, bmin, bmax)
t = time.time(); g.build((sa.MODE_BVH,)); print('host SAH build+upload s', time.time() - t)
t = time.time(); g.build((sa.MODE_BVH,), on_device=True); print('device LBVH build s (incl. v9/slab upload)', time.time() - t)
t = time.time(); g.build((sa.MODE_BVH,), on_device=True); print('device LBVH build again s', time.time() - t)
