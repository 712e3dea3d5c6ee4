"""How many frames really run concurrently: raw HIP streams (ctypes) vs torch streams, with and without priorities."""
import sys, time, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import workloads
rt.init(0)
hip = C.CDLL("libamdhip64.so")
cfg = workloads.default_config()
scene = rt.Scene(workloads.bvh_stress_scene())
dev = torch.device("cuda", 0)
lo, hi = C.c_int(0), C.c_int(0)
hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
print("priority range", lo.value, hi.value, "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
def raw_streams(n, prios):
    out = []
    for i in range(n):
        s = C.c_void_p()
        assert hip.hipStreamCreateWithPriority(C.byref(s), 1, prios[i % len(prios)]) == 0
        out.append(s)
    return out
bufs = [torch.empty(1080 * 1920 * 3, dtype=torch.uint8, device=dev) for _ in range(8)]
p = cfg.params(rng="d0"); p.tile_rows, p.part_index, p.part_count = 8, 0, 8
for name, prios in (("same priority", [0]), ("3 priorities", [lo.value, (lo.value + hi.value) // 2, hi.value])):
    streams = raw_streams(8, prios)
    for F in (1, 2, 3, 4, 6, 8):
        def run(K):
            for i in range(K):
                scene.render_device(p, bufs[i % F].data_ptr(), stream=streams[i % F])
        run(2 * F); torch.cuda.synchronize()
        K = 48
        t0 = time.perf_counter(); run(K); torch.cuda.synchronize()
        print("%s: in flight %d: %.3f ms/step" % (name, F, (time.perf_counter() - t0) * 1e3 / K), flush=True)
