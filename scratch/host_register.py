import ctypes as C, numpy as np, mmap
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipGetErrorString.restype = C.c_char_p
hip.hipInit(0)
def reg(ptr, n, flags, what):
    rc = hip.hipHostRegister(C.c_void_p(ptr), C.c_size_t(n), C.c_uint(flags))
    print("%-40s flags %d -> %d %s" % (what, flags, rc, hip.hipGetErrorString(rc).decode()))
    if rc == 0: hip.hipHostUnregister(C.c_void_p(ptr))
    else: hip.hipGetLastError()
n = 8 << 20
a = np.zeros(n, dtype=np.uint8)
for fl in (0, 1, 2):
    reg(a.ctypes.data, n, fl, "numpy 8 MB")
m = mmap.mmap(-1, n)
buf = (C.c_char * n).from_buffer(m)
reg(C.addressof(buf), n, 0, "anonymous mmap, page aligned")
pa = (a.ctypes.data + 4095) & ~4095
reg(pa, n - 8192, 0, "numpy, page-aligned sub-range")
import resource
print("RLIMIT_MEMLOCK", resource.getrlimit(resource.RLIMIT_MEMLOCK))
