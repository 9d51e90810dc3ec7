import sys, os, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import softray_amd as sa
from softray_amd.renderer import Renderer, Instance, Vector
r = Renderer(); r.BackgroundColor = 0xff00ff
px = np.zeros(1024 * 1024, dtype=np.int32); r.SetRenderingSurface(1024, 1024, px)
with open(os.path.join(ROOT, "tests", "golden", "obj.3ds"), "rb") as f: r.Load3dsModelFromStream(f)
inst = Instance(r.Model, Position=Vector(0, 0, 1.0), Yaw=135 / 180 * math.pi, Pitch=-22 / 180 * math.pi); r.Instances.append(inst)
r.rayTrace = True; r.gpuTraceMode = sa.MODE_BVH; r.rayTraceShadows = True; r.rayTraceFocalBlur = False
r.Render()
f = r.BuildFrame(inst)
buf = torch.empty(1024 * 1024, dtype=torch.int32, device="cuda")
g = r._scene
for nosplit in (0, 1):
    ff = sa.Frame.from_buffer_copy(bytes(f))
    if nosplit: ff.flags |= sa._lib.F_NO_SPLIT
    s = torch.cuda.current_stream()
    for _ in range(3): g.render_device(ff, buf.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize(); g.reset_kernel_times()
    t = time.perf_counter()
    for _ in range(10): g.render_device(ff, buf.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10 * 1e3
    print("nosplit", nosplit, round(dt, 3), {k: (round(v[0] / 10, 3), v[1] // 10) for k, v in g.kernel_times().items()}, g.debug_counters())
for sp in ("2", "1"):
    os.environ["SR_SPLIT"] = sp
    for _ in range(3): r.Render()
    t = time.perf_counter()
    for _ in range(10): r.Render()
    print("Render() SR_SPLIT", sp, round((time.perf_counter() - t) / 10 * 1e3, 3))
    g.reset_kernel_times(); r.Render(); print({k: (round(v[0], 3), v[1]) for k, v in g.kernel_times().items()})
