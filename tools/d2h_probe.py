"""How long 4 MB take from device to host: pageable numpy, torch-pinned, hipHostMalloc'ed, and a host memcpy of the same size.
    python tools/d2h_probe.py"""
import ctypes as C, time, sys
import numpy as np, torch
hip = C.CDLL("libamdhip64.so")
dev = torch.device('cuda', 0)
n = 1024 * 100 * 40 + 4096
src = torch.zeros(n, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
def t(fn, reps=50):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
pg = np.zeros(n, np.uint8)
print("hipMemcpy D2H to pageable numpy: %.0f us" % t(lambda: hip.hipMemcpy(pg.ctypes.data, src.data_ptr(), n, 2)))
tp = torch.empty(n, dtype=torch.uint8).pin_memory()
print("hipMemcpy D2H to torch-pinned: %.0f us" % t(lambda: hip.hipMemcpy(tp.data_ptr(), src.data_ptr(), n, 2)))
p = C.c_void_p()
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
for flags, name in ((0, "default"), (0x40000000, "non-coherent"), (0x80000000, "coherent")):
    p = C.c_void_p()
    rc = hip.hipHostMalloc(C.byref(p), n, flags)
    if rc: print(name, "hipHostMalloc failed", rc); continue
    print("hipMemcpy D2H to hipHostMalloc(%s): %.0f us" % (name, t(lambda: hip.hipMemcpy(p.value, src.data_ptr(), n, 2))))
    dst = np.zeros(n, np.uint8)
    print("   host memcpy from it to pageable: %.0f us" % t(lambda: C.memmove(dst.ctypes.data, p.value, n)))
a = np.zeros(n, np.uint8); b = np.ones(n, np.uint8)
print("host memcpy pageable -> pageable: %.0f us" % t(lambda: C.memmove(a.ctypes.data, b.ctypes.data, n)))
