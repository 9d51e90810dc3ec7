"""How should a blocking sr_render hand a 64 MiB frame to a caller's PAGEABLE buffer?  Measures, on this box:
pageable hipMemcpy D2H, pinned D2H, hipHostRegister + async D2H + hipHostUnregister of the pageable buffer, and the CPU
memcpy from a pinned staging buffer (1 thread / 4 threads).  usage: python scripts/gpu_hostcopy.py [MiB]"""
import ctypes as C, json, sys, time, threading
import numpy as np, torch
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = mib << 20
hip = C.CDLL("libamdhip64.so")
dev = torch.empty(n, dtype=torch.uint8, device="cuda").random_()
page = np.zeros(n, dtype=np.uint8)
pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
torch.cuda.synchronize()
out = {"MiB": mib}
def best(fn, reps=5):
    b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t)
    return b * 1e3
vp = C.c_void_p
out["pageable_d2h_ms"] = best(lambda: hip.hipMemcpy(vp(page.ctypes.data), vp(dev.data_ptr()), C.c_size_t(n), 2))
out["pinned_d2h_ms"] = best(lambda: pinned.copy_(dev))
def reg_copy():
    t0 = time.perf_counter()
    rc = hip.hipHostRegister(vp(page.ctypes.data), C.c_size_t(n), 0)
    t1 = time.perf_counter()
    assert rc == 0, rc
    hip.hipMemcpyAsync(vp(page.ctypes.data), vp(dev.data_ptr()), C.c_size_t(n), 2, None)
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    hip.hipHostUnregister(vp(page.ctypes.data))
    t3 = time.perf_counter()
    reg_copy.parts = ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3)
out["register_copy_unregister_ms"] = best(reg_copy)
out["register_ms, copy_ms, unregister_ms"] = reg_copy.parts
pn = pinned.numpy()
out["cpu_memcpy_1thread_ms"] = best(lambda: np.copyto(page, pn))
def par(k):
    th = [threading.Thread(target=lambda i=i: np.copyto(page[i * n // k:(i + 1) * n // k], pn[i * n // k:(i + 1) * n // k])) for i in range(k)]
    [t.start() for t in th]; [t.join() for t in th]
out["cpu_memcpy_4threads_ms"] = best(lambda: par(4))
out["cpu_memcpy_8threads_ms"] = best(lambda: par(8))
print(json.dumps(out))
