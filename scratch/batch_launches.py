"""scratch/batch_launches.py [frames]: nothing but launches of `frames` (default 16) C2 frames per call, one at a time (the
dominant launch of bench.py's timed region), for rocprofv3 --pmc / --kernel-trace: every k_traverse dispatch of this
process is such a launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import workloads
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rt.init(0)
dev = torch.device("cuda", 0)
scene = rt.Scene(workloads.bvh_stress_scene(1_000_000, 1920, 1080))
p = workloads.default_config().params(rng=os.environ.get("RT_BATCH_RNG", "seeded"), seed=12345)   # the sampling of bench.py's timed region
if os.environ.get("RT_BATCH_SCHEDULE") == "fp32":   # the throughput variant
    p.schedule = rt.RT_SCHEDULE_WAVEFRONT_FP32
buf = torch.zeros((F, 1080, 1920, 3), dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(dev)
torch.cuda.synchronize()
for _ in range(3):
    scene.render_frames_device(p, F, buf.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
ms, n = scene.traverse_time(st.cuda_stream)
print("k_traverse: %d launches of %d frames, %.3f ms each" % (n, F, ms / max(1, n)))
