"""What a fresh hipMalloc costs on this box: per call and per byte (the cold constructor's 16-20 ms)."""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
def t_alloc(sizes):
    ps = []
    t0 = time.perf_counter()
    for s in sizes:
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), s) == 0
        ps.append(p)
    dt = 1e3 * (time.perf_counter() - t0)
    return dt, ps
p0 = C.c_void_p(); hip.hipMalloc(C.byref(p0), 1024)      # (runtime initialised)
MB = 1 << 20
for label, sizes in (("35 x 2 MB", [2 * MB] * 35), ("1 x 70 MB", [70 * MB]), ("4 x 100 MB", [100 * MB] * 4), ("1 x 400 MB", [400 * MB]),
                     ("35 x 2 MB again", [2 * MB] * 35), ("35 x 64 KB", [64 << 10] * 35)):
    dt, ps = t_alloc(sizes)
    print(f"{label:18s} {dt:7.2f} ms")
    if "again" not in label and label != "35 x 2 MB":
        t0 = time.perf_counter()
        for p in ps: hip.hipFree(p)
        print(f"   free            {1e3 * (time.perf_counter() - t0):7.2f} ms")
dt, ps = t_alloc([400 * MB])
print(f"1 x 400 MB after a free of the same {dt:7.2f} ms")
