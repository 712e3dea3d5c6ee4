import time, torch, numpy as np, ctypes
dev = torch.device("cuda", 0)
for mb in (6, 25, 100):
    n = mb << 20
    d = torch.zeros(n, dtype=torch.uint8, device=dev)
    pinned = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    pageable = torch.empty(n, dtype=torch.uint8)
    reg = np.zeros(n, dtype=np.uint8)
    rt = torch.cuda.cudart()
    rc = rt.cudaHostRegister(reg.ctypes.data, n, 0)
    regt = torch.from_numpy(reg)
    for name, h in (("pinned (hipHostMalloc)", pinned), ("pageable", pageable), ("hipHostRegister of numpy memory (rc %s)" % (rc,), regt)):
        for _ in range(3): h.copy_(d, non_blocking=True); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10): h.copy_(d, non_blocking=True); torch.cuda.synchronize()
        ms = (time.perf_counter() - t) * 100
        print("%4d MB  %-50s %.3f ms  %.1f GB/s" % (mb, name, ms, n / ms / 1e6))
    rt.cudaHostUnregister(reg.ctypes.data)
