#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X render core (contract: see the task's driver).

Metric (BASELINE.json): Mrays/s (primary + secondary) and ms/frame, 1080p Whitted, 1/2/4/8 GPUs.
Workload: BASELINE config 2 — 10^6 seeded random ellipsoids (raytracer_amd.workloads.bvh_stress_scene), 1920x1080,
1 spp, default flags, shipped config values; the reference's sampling (a random position inside the pixel for every ray, seeded
per frame; --sampling d0 selects the deterministic mode of the parity tests).

A "step" is one whole frame: every rank renders its interleaved row tiles with the HIP kernels, 16 frames per GPU of the job per call
(C-ABI rt_render_frames_device: the rays of the frames share the ray queues and one launch of every kernel; two such
calls in flight on HIP streams of their own), and the u8 slices of those frames travel to rank 0 inside the library in
one exchange (rt_gather_frames: RCCL send/recv + k_deinterleave).  The scene is resident in HBM when the timed region starts and the
frame stays in HBM.  value = rays of all ranks / max-over-ranks time.  Scaling is "strong": the frame is fixed as N grows.
`python bench.py --gpus N` without a launcher's environment starts its N ranks itself (torch.distributed.run) and fails
when fewer GPUs are visible; it never prints an N=1 line for N>1.

Rank 0 prints ONE JSON line.  Extra objects: "roofline" of k_traverse -- bound "valu": useful lane-operations per second
(VALU wave-instructions x 64 x active-lane fraction from the committed PMC passes, profiles/pmc_c2.json, over the kernel
time measured live with HIP events) against 39.32 Tlane-op/s; "traffic" = measured HBM bytes per launch; the SURVEY 8(d)
node bytes per second under both per-sphere constants -- then "cpu_baseline" (the CPU oracle, OpenMP, on this host's
cores, checked against the real reference binary where that exists; N=1 only) and "configs" (the other BASELINE
configurations timed whole-frame, with the kernel that dominates each from profiles/config_kernels.json).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP maps streams onto at most GPU_MAX_HW_QUEUES hardware queues (4 by default) and kernels of streams that share a
# queue run in order; the frames in flight below want one queue each (plus torch's and RCCL's own streams).  Read by the HIP runtime when it starts.
try:
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 16:
        os.environ["GPU_MAX_HW_QUEUES"] = "16"
except ValueError:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"
# Throughput setting of the library: traversal grids of at least 8 rays per lane, so that the launches of the frames in
# flight fit the GPU side by side when a rank's share of the frame is small (the default, 2, favours one frame's latency).
os.environ.setdefault("RT_RAYS_PER_LANE", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import raytracer_amd as rt  # noqa: E402
from raytracer_amd import workloads  # noqa: E402
from raytracer_amd.distributed import FrameGather, init_comm  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md, L2 (per XCD): 4 MiB per XCD, ~34.5 TB/s aggregate
NODE_BYTES = 64           # one DNode4: four quantised child boxes + child references (device_types.h)
SPHERE_BYTES_LAYOUT = 16 + 144   # what a sphere test really reads here: DPrim header + FP64 DXform (device_types.h)
SPHERE_BYTES_SURVEY = 64         # SURVEY 8(d)'s constant (3x4 FP32 inverse affine + ids)
RAY_STATE_BYTES = 16      # amortised framebuffer + ray state traffic per ray
# VALU peak of the chip in lane-operations: 256 CUs x 4 SIMDs x 16 lanes per clock x 2.4 GHz (MI355X_MICROARCH.md); one
# wave instruction of 64 lanes takes 4 cycles on a SIMD whatever its lanes do, so (instructions x active lanes) per second
# against this figure is issue fraction x active-lane fraction: the share of the machine's lane-cycles doing work.
VALU_PEAK_TLANEOPS = 256 * 4 * 16 * 2.4e9 / 1e12
FLOPS_PER_BOX_TEST, FLOPS_PER_SPHERE_TEST = 22, 60
VALU_PEAK_FP32_TFLOPS = 157.3


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--spheres", type=int, default=1_000_000, help="C2 primitive count (default = the named config)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tile-rows", type=int, default=8)
    ap.add_argument("--sampling", default="seeded", choices=["seeded", "d0"],
                    help="seeded: the reference's sampling (main.cpp:288-302: a random position inside the pixel for every ray; frame f of a "
                         "launch uses seed + f); d0: the deterministic mode of the parity tests (pixel centres, all frames equal)")
    ap.add_argument("--frames-per-launch", type=int, default=None,
                    help="frames rendered per call (rt_render_frames_device) and sent to rank 0 per exchange; default 16 per GPU of the job, "
                         "at most 64 (a rank's share of a frame shrinks with the job, so the launch keeps its size)")
    ap.add_argument("--sets", type=int, default=2, help="slice buffers that alternate, so that the exchange of one launch's frames overlaps the next launch "
                                                         "(all launches are issued on ONE stream: two traversal launches side by side were measured slower)")
    ap.add_argument("--frames-in-flight", type=int, default=None,
                    help="the round-1 setting instead: ONE frame per call (rt_render_device), this many in flight on streams of their own")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the timings of the other BASELINE configs (C1, C3, C4, C5 share)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL over xGMI; the measured configuration) or gloo "
                    "(rehearsal of the multi-rank control flow with the ranks sharing the visible GPUs; slices are staged through the host)")
    ap.add_argument("--cpu-row-step", type=int, default=1, help="CPU baseline renders every k-th row of the same frame")
    ap.add_argument("--cpu-passes", type=int, default=18, help="CPU baseline: frames rendered (sized for ~10 s of CPU work on 16 threads)")
    return ap.parse_args()


def cpu_baseline(scene_desc, params, row_step, passes):
    """Oracle (port of the reference algorithm, oracle/rt_oracle.cpp) timed on this host: pixel loop only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    h = scene_desc.height
    secs, rays = 0.0, 0
    for _ in range(max(1, passes)):
        _, cnt = oracle_py.render(scene_desc, params, threads=threads, counters=True, rows=(0, h, row_step))
        secs += oracle_py.last_render_seconds()
        rays += cnt[0]
    try:
        check = reference_check(threads)
    except Exception as e:   # the baseline itself does not depend on it
        check = {"error": str(e)[:200]}
    return {
        "value": round(rays / secs / 1e6, 4), "unit": "Mrays/s", "cores": oracle_py.last_threads(), "kind": "port",
        "reference_check": check,
        "sample": "same 10^6-sphere 1080p frame, every %d-th row, %d passes (%d rays), pixel loop only (BVH build excluded), %.2f s on %d threads"
                  % (row_step, passes, rays, secs, oracle_py.last_threads()),
        "box_tests_per_ray": round(cnt[1] / cnt[0], 2), "prim_tests_per_ray": round(cnt[2] / cnt[0], 2),
    }


def reference_check(threads, n_spheres=100_000):
    """The REAL reference binary (oracle/_ref/ref_raytracer: the reference's own sources compiled in place by
    oracle/Makefile, with only its RNG replaced by the deterministic shim) next to the port on an input it loads in
    seconds: the same C2 frame with 10^5 spheres (at 10^6 its BVH build alone takes 32 s).  Render-only time of the
    reference = wall time of the 1080p run minus that of a 16x9 run of the same scene (load + build), SURVEY 8(d)."""
    import oracle_py
    import run_ref
    if not os.path.exists(run_ref.REF_BIN):
        return None
    text = workloads.bvh_stress_scene_text(n_spheres, 1920, 1080)
    cfg_text = open(workloads.DEFAULT_CONFIG_PATH).read()
    t0 = time.perf_counter()
    run_ref.run_reference(run_ref.override_resolution(text, 16, 9), cfg_text, flags=["--parallel"], rng="d0", threads=threads)
    t1 = time.perf_counter()
    ref_img, _ = run_ref.run_reference(text, cfg_text, flags=["--parallel"], rng="d0", threads=threads)
    t2 = time.perf_counter()
    ref_secs = max(1e-3, (t2 - t1) - (t1 - t0))
    cfg = rt.Config(text=cfg_text)
    scene_file = rt.SceneFile(text=text, texture_root=ROOT, config=cfg)
    port_img, cnt = oracle_py.render(scene_file, cfg.params(rng="d0"), threads=threads, counters=True)
    port_secs = oracle_py.last_render_seconds()
    return {"workload": "C2 with %d spheres, 1920x1080, 1 spp (scene text parsed by both)" % n_spheres, "cores": threads,
            "reference_mrays_per_s": round(cnt[0] / ref_secs / 1e6, 4), "port_mrays_per_s": round(cnt[0] / port_secs / 1e6, 4),
            "port_over_reference": round(ref_secs / port_secs, 3), "reference_render_s": round(ref_secs, 3),
            "images_identical": bool(np.array_equal(ref_img, port_img))}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (torch.distributed.run, one process per GPU)
    BEFORE anything in this process touches a GPU, and pass on their exit code.  Never a single-rank result for N > 1."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def config_cpu_baseline(scene_file, params, rays_per_frame, height):
    """The oracle port on this host's cores on a row subset of the same frame (every k-th row, k chosen so that the sample is
    about 4 x 10^7 rays: a few seconds), as a rate: rows are independent, so the rate of the sample is the rate of the frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    step = max(1, int(np.ceil(rays_per_frame / 4e7)))
    secs, rays, passes = 0.0, 0, 0
    try:
        while secs < 2.0 and passes < 200:   # small frames take milliseconds: repeat until the sample is a couple of seconds
            _, cnt = oracle_py.render(scene_file, params, threads=threads, counters=True, rows=(0, height, step))
            secs += oracle_py.last_render_seconds()
            rays += cnt[0]
            passes += 1
    except Exception as e:   # noqa: BLE001
        return {"error": str(e)[:200]}
    return {"value": round(rays / secs / 1e6, 4), "unit": "Mrays/s", "cores": oracle_py.last_threads(), "kind": "port",
            "sample": "every %d-th row of the same frame, %d passes (%d rays, %.2f s on %d threads); pixel loop only" % (step, passes, rays, secs, oracle_py.last_threads())}


def other_configs(dev, with_cpu=True):
    """The other BASELINE configs on this GPU (N = 1 only, after the headline): whole frames from the committed fixtures at the
    configs' sizes, scene resident, output in HBM; rays from an untimed counting pass.  The kernel that dominates each frame
    and its share come from the committed kernel traces (profiles/config_kernels.json), labelled as such."""
    import tempfile
    shares = {}
    sp = os.path.join(ROOT, "profiles", "config_kernels.json")
    if os.path.exists(sp):
        try:
            shares = json.load(open(sp))
        except Exception:
            shares = {}
    res = []
    stream = torch.cuda.Stream(dev)
    with tempfile.TemporaryDirectory(prefix="rt_bench_") as scratch:
        for label, name, w, h, spp, rng in workloads.BASELINE_CONFIGS:
            try:
                sf, p = workloads.golden_scene(name, w, h, spp, rng, sky_dir=scratch)
                sc = rt.Scene(sf)
                _, st = sc.render(p, stats=True)
                buf = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
                torch.cuda.synchronize(dev)
                sc.render_device(p, buf.data_ptr(), stream=stream.cuda_stream)
                torch.cuda.synchronize(dev)
                reps = 2 if st.rays > 2e8 else 10
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(stream):
                    e0.record(stream)
                    for _ in range(reps):
                        sc.render_device(p, buf.data_ptr(), stream=stream.cuda_stream)
                    e1.record(stream)
                torch.cuda.synchronize(dev)
                ms = e0.elapsed_time(e1) / reps
                ms16 = None
                if spp == 1:   # a 1-spp frame is a handful of small launches: 16 frames (seeds seed + f) per call share them
                    buf16 = torch.empty((16, h, w, 3), dtype=torch.uint8, device=dev)
                    sc.render_frames_device(p, 16, buf16.data_ptr(), stream=stream.cuda_stream)
                    torch.cuda.synchronize(dev)
                    with torch.cuda.stream(stream):
                        e0.record(stream)
                        for _ in range(2):
                            sc.render_frames_device(p, 16, buf16.data_ptr(), stream=stream.cuda_stream)
                        e1.record(stream)
                    torch.cuda.synchronize(dev)
                    ms16 = e0.elapsed_time(e1) / 32
                    del buf16
                ms_acc = None
                if rng == "seeded" and spp > 1 and w * h >= 3840 * 2160:
                    # config 5 proper is 64 such passes added up in HBM (rt_accumulate_device: one pass after the other, chunks two at a time)
                    d_sum = torch.zeros((h, w, 3), dtype=torch.float64, device=dev)
                    sc.accumulate_device(p, d_sum.data_ptr(), 0, 2, stream=stream.cuda_stream)
                    torch.cuda.synchronize(dev)
                    t_acc = time.perf_counter()
                    sc.accumulate_device(p, d_sum.data_ptr(), 2, 4, stream=stream.cuda_stream)
                    torch.cuda.synchronize(dev)
                    ms_acc = (time.perf_counter() - t_acc) * 1e3 / 4
                    del d_sum
                sc.close()
                k = shares.get(name, {})
                traced = int(st.rays) - int(st.rays_coincident)
                cpu = config_cpu_baseline(sf, p, int(st.rays), h) if with_cpu else None
                res.append({"config": label, "fixture": name, "resolution": [w, h], "spp": spp, "rays_per_frame": int(st.rays),
                            # rays as the reference counts them (its world.intersect calls); of these, the 2 x shadow_samples rays of a hit
                            # towards a light of radius 0 are one and the same ray, which is traced once: both counts, both rates
                            "rays_traced_per_frame": traced,
                            "ms": round(ms, 3), "Mrays_per_s": round(st.rays / ms / 1e3, 1), "Mrays_per_s_traced": round(traced / ms / 1e3, 1),
                            "frames_timed": reps,
                            "ms_per_frame_16_frames_per_call": round(ms16, 3) if ms16 else None, "sampling": rng,
                            "ms_per_pass_in_rt_accumulate_device": round(ms_acc, 3) if ms_acc else None,
                            "dominant_kernel": k.get("kernel"), "dominant_kernel_fraction": k.get("fraction"),
                            "dominant_kernel_source": k.get("source")})
                if cpu:
                    cpu["gpu_over_cpu"] = round((st.rays / ms / 1e3) / cpu["value"], 1) if cpu["value"] > 0 else None
                    res[-1]["cpu_baseline"] = cpu
                if k.get("valu"):
                    # SURVEY 8(d), small scenes: bound by VALU issue under divergence.  Roofline of the frame's kernels from the committed
                    # PMC passes (instructions and active lanes are fixed by the build and the frame; the busy time was measured in the
                    # same passes): achieved = VALU wave-instructions x 64 x active-lane fraction / GPU busy time, peak = 39.32 Tlane-op/s.
                    v = k["valu"]
                    res[-1]["roofline"] = {"bound": "valu", "kernel": k.get("kernel"), "unit": "Tlane-op/s", "peak": round(VALU_PEAK_TLANEOPS, 2),
                                           "achieved": round(v["issue_fraction"] * v["active_lane_fraction"] * VALU_PEAK_TLANEOPS, 3),
                                           "frac": round(v["issue_fraction"] * v["active_lane_fraction"], 4),
                                           "issue_fraction": v["issue_fraction"], "active_lane_fraction": v["active_lane_fraction"],
                                           "kernels": v.get("kernels"), "run": v.get("run"), "traffic": v.get("hbm_bytes_per_frame"),
                                           "pmc_source": v.get("source")}
            except Exception as e:   # the headline does not depend on these
                res.append({"config": label, "fixture": name, "error": str(e)[:200]})
        rt.release_workspaces()
    return res


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.backend == "nccl" and torch.cuda.device_count() < args.gpus:   # counting devices does not initialise the GPU
            raise SystemExit("bench.py: --gpus %d asked for, %d visible: not measured" % (args.gpus, torch.cuda.device_count()))
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())   # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    rt.init(local_rank)
    native_exchange = None
    if world > 1 and args.backend == "nccl":
        # the library's own RCCL communicator: slices travel by rt_gather_frames (comm.hip).  Should it fail to come up on any
        # rank, every rank falls back to torch.distributed's gather (RCCL too) + index_select, and the line says so.
        ok = 1
        try:
            init_comm(dev)
        except Exception as e:   # noqa: BLE001
            ok = 0
            print("bench.py: rank %d: the library's communicator did not come up (%s)" % (rank, str(e)[:200]), file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        native_exchange = bool(flag.item())

    cfg = workloads.default_config()
    scene_desc = workloads.bvh_stress_scene(args.spheres, args.width, args.height)
    params = cfg.params(rng=args.sampling, seed=12345)
    params.tile_rows, params.part_index, params.part_count = args.tile_rows, rank, world
    t_create = time.perf_counter()
    scene = rt.Scene(scene_desc)  # BVH build + upload: outside the timed region (scene resident in HBM)
    t_create = time.perf_counter() - t_create
    W, H = args.width, args.height
    # Frames are independent and a frame (or a rank's share of one) is a small launch whose persistent traversal kernel
    # ends with a tail of its longest rays, so B frames are rendered per call (rt_render_frames_device: the rays of the B
    # frames share the queues and one tail) into the B slices of a set, and the set travels to rank 0 in ONE exchange
    # (rt_gather_frames: at 8 ranks a rank's share of a frame is 0.78 MB, so a message per frame would be bound by its launch
    # and host cost).  All launches go to one HIP stream, back to back (the call is asynchronous for this frame: nothing is
    # read back); S sets of slices alternate, so that the exchange of one launch's frames, on a stream of its own, overlaps
    # the next launch.  This one thread issues everything in step order on every rank.
    # --frames-in-flight F selects the round-1 setting instead: one frame per call on F streams.
    B = max(1, args.frames_per_launch) if args.frames_per_launch else min(64, 16 * world)
    B_steady = B
    # never more frames per launch than the timed region has (a set would ship empty slots), and with several GPUs at least two
    # batches, so that one batch's exchange overlaps the other's render
    B = min(B, max(1, args.steps))
    if world > 1 and args.steps >= 2:
        B = min(B, (args.steps + 1) // 2)
    S, n_streams = max(1, args.sets), 1
    if args.frames_in_flight is not None:
        B, S = 1, max(1, args.frames_in_flight)
        n_streams = S
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    counters = torch.zeros(8, dtype=torch.int64, device=dev)   # RT_N_COUNTERS
    sets = [FrameGather(W, H, args.tile_rows, dev, frames=B, native=native_exchange) for _ in range(S)]   # slices [B, max_rows, W, 3]; frames on rank 0
    batch_events, gather_events = [], []

    # All exchanges are issued on ONE stream of their own, in batch order (operations of a communicator stay ordered whatever
    # the RCCL version makes of several user streams), tied to the render streams by events both ways: the exchange of a set
    # waits for the set's render, and the next render into the set waits until its previous exchange has read the slices.
    gather_stream = torch.cuda.Stream(dev) if world > 1 else None
    done = [None] * S

    def launch(k, nb=None, collect=False, timed=False):
        """Batch k: nb (default B) frames into set k % S and the set's exchange.  Collective: every rank issues the same batches."""
        stream, fg = streams[k % n_streams], sets[k % S]
        with torch.cuda.stream(stream):
            if done[k % S] is not None:
                stream.wait_event(done[k % S])
            if timed:
                batch_events.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), nb or B))
                batch_events[-1][0].record(stream)
            if collect or B == 1:
                scene.render_device(params, fg.slices.data_ptr(), counters.data_ptr() if collect else None, collect_counters=collect,
                                    stream=stream.cuda_stream)
            else:
                scene.render_frames_device(params, nb or B, fg.slices.data_ptr(), slice_rows=fg.max_rows, stream=stream.cuda_stream)
            if timed:
                batch_events[-1][1].record(stream)
            if world > 1:
                rendered = torch.cuda.Event()
                rendered.record(stream)
        if world > 1:
            with torch.cuda.stream(gather_stream):
                gather_stream.wait_event(rendered)
                if timed:
                    gather_events.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
                    gather_events[-1][0].record(gather_stream)
                fg.run(1 if (collect or B == 1) else (nb or B))   # RCCL send/recv of the batch's slices to rank 0 + k_deinterleave: a partial batch ships its own bytes
                if timed:
                    gather_events[-1][1].record(gather_stream)
                done[k % S] = torch.cuda.Event()
                done[k % S].record(gather_stream)

    def run(frames, timed=False, first_batch=0):
        i, k = 0, first_batch
        while i < frames:
            nb = min(B, frames - i)
            launch(k, nb, timed=timed)
            i, k = i + nb, k + 1
        return k

    # one untimed counting pass: rays, box tests, primitive tests per frame of this rank
    counters.zero_()
    launch(0, collect=True)
    torch.cuda.synchronize(dev)
    c = counters.cpu().numpy().astype(np.int64)
    if world > 1:
        tot = torch.from_numpy(c).to(dev)
        dist.all_reduce(tot)
        c_all = tot.cpu().numpy()
    else:
        c_all = c
    rays_frame, boxes_frame, prims_frame, nodes_frame = int(c_all[0]), int(c_all[1]), int(c_all[2]), int(c_all[4])
    rays_rank, prims_rank, nodes_rank = int(c[0]), int(c[2]), int(c[4])

    # warm-up: at least W frames and at least one batch per set (every stream allocates its queues, RCCL warms up)
    k0 = run(max(args.warmup, B * S))
    torch.cuda.synchronize(dev)
    for st in streams:
        scene.traverse_time(st.cuda_stream)   # drop the warm-up launches from the kernel timing
    if world > 1:
        rt.gather_time()                      # ... and the warm-up gathers from the exchange timing
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(args.steps, timed=True, first_batch=k0)
    t_enqueued = time.perf_counter() - t0   # host time to enqueue all steps (launches + exchanges), before waiting for the GPU
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # time from the first kernel of a frame's batch to its last (the frame is complete when its batch is)
    render_ms = float(np.mean([a.elapsed_time(b) for a, b, _ in batch_events])) if batch_events else 0.0
    # Where the timed region went, per rank: device time of the renders, of the gathers (as the gather stream saw them: waiting for
    # the other ranks' slices included) and, inside the library's own exchange, of its two halves (rt_gather_time: grouped
    # ncclSend / ncclRecv, then k_deinterleave on rank 0).  The line carries the maximum over the ranks and rank 0's own figures:
    # rank 0 is receiver, de-interleaver and a full renderer at once, so an imbalance shows here and not as a slow render.
    phases = None
    if world > 1:
        mine = [sum(a.elapsed_time(b) for a, b, _ in batch_events), sum(a.elapsed_time(b) for a, b in gather_events), 0.0, 0.0]
        if native_exchange:
            ex_ms, de_ms, _n = rt.gather_time()
            mine[2], mine[3] = ex_ms, de_ms
        t = torch.tensor(mine, dtype=torch.float64, device=dev)
        tm = t.clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        t0_ = t.clone()
        dist.broadcast(t0_, src=0)
        keys = ["render_ms", "gather_ms", "exchange_ms", "deinterleave_ms"]
        phases = {"max_over_ranks": {k: round(float(v), 4) for k, v in zip(keys, tm.cpu().tolist())},
                  "rank0": {k: round(float(v), 4) for k, v in zip(keys, t0_.cpu().tolist())},
                  "timed_region_ms": round(elapsed * 1e3, 4), "batches": len(batch_events),
                  "bytes_per_exchange_per_rank": [int(nb * sets[0].max_rows * W * 3) for _, _, nb in batch_events],
                  "note": "device time summed over the timed region's batches; render and gather run on streams of their own and overlap"}
    # The driver's --steps is small against a launch (20 frames over 8 GPUs are 2.5 frames of rays per rank and launch: one tail of
    # the persistent traversal kernel per launch, and an exchange with nothing left to hide behind): the same job over a region
    # long enough to fill the launches, for diagnosis -- NOT the value of this line.
    steady = None
    if world > 1 and args.frames_in_flight is None and B_steady > B:
        big = [FrameGather(W, H, args.tile_rows, dev, frames=B_steady, native=native_exchange) for _ in range(S)]
        saved = (sets, B)
        sets, B = big, B_steady
        done[:] = [None] * S
        run(B_steady * S)                     # warm-up: every set once
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        ts = time.perf_counter()
        n_steady = 4 * B_steady
        run(n_steady)
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        tmax2 = torch.tensor([time.perf_counter() - ts], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax2, op=dist.ReduceOp.MAX)
        steady = {"steps": n_steady, "frames_per_launch": B_steady, "ms_per_step": round(float(tmax2.item()) * 1e3 / n_steady, 4),
                  "Mrays_per_s": round(rays_frame * n_steady / float(tmax2.item()) / 1e6, 1),
                  "note": "same job, same barriers, %d steps in launches of %d frames: what the partition sustains; not this line's value" % (n_steady, B_steady)}
        sets, B = saved
        done[:] = [None] * S
        del big
    # device time of the k_traverse launches of the timed region (hipEvents recorded by the library on the launch streams):
    # launches in flight on several streams overlap and share the GPU, so each one lasts longer than it would alone
    tt = [scene.traverse_time(st.cuda_stream) for st in streams]
    launches = sum(n for _, n in tt)
    traverse_ms_in_flight = sum(ms for ms, _ in tt) / max(1, launches)
    # roofline pass: the same batches one at a time on one stream (no barrier needed: ranks render independently), so that a
    # launch's duration is the kernel's own; this is what `rocprofv3 --kernel-trace --stats -- python3 bench.py
    # ` reports too
    for i in range(3 if B > 1 else max(3, min(args.steps, 10))):
        launch(0)
        torch.cuda.synchronize(dev)
    ms, launches_alone = scene.traverse_time(streams[0].cuda_stream)
    traverse_ms = ms / max(1, launches_alone)
    if world > 1:
        dist.barrier()
    # outside the timed region: the frames assembled from the ranks' slices against the same frame rendered whole on rank 0
    frame_ok = None
    if world > 1:
        launch(0)
        torch.cuda.synchronize(dev)
        if rank == 0:
            whole = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
            p1 = params.copy()
            p1.part_count, p1.part_index = 1, 0
            frame_ok = True
            for b in range(B):   # frame b of a launch is the frame rendered alone with seed + b
                p1.seed = params.seed + (b if B > 1 else 0)
                scene.render_device(p1, whole.data_ptr(), stream=streams[0].cuda_stream)
                torch.cuda.synchronize(dev)
                frame_ok = frame_ok and bool(torch.equal(whole, sets[0].frames[b]))
        dist.barrier()

    # one frame at a time (latency setting) and the boundary's own contract -- rt_render into a HOST buffer (main.cpp:242,320)
    solo_ms, host_ms, traverse_ms_one_frame, normals_ms, fp32_variant, host_frames_ms = None, None, None, None, None, {}
    if world == 1:
        torch.cuda.synchronize(dev)
        n_solo = max(5, min(args.steps, 20))
        scene.render_device(params, sets[0].slices.data_ptr(), stream=streams[0].cuda_stream)
        torch.cuda.synchronize(dev)
        scene.traverse_time(streams[0].cuda_stream)
        t1 = time.perf_counter()
        for _ in range(n_solo):
            scene.render_device(params, sets[0].slices.data_ptr(), stream=streams[0].cuda_stream)
            torch.cuda.synchronize(dev)
        solo_ms = (time.perf_counter() - t1) * 1e3 / n_solo
        ms, n1 = scene.traverse_time(streams[0].cuda_stream)
        traverse_ms_one_frame = ms / max(1, n1)
        p_host = params.copy()
        p_host.part_count, p_host.part_index = 1, 0
        host_buf = np.zeros((H, W, 3), dtype=np.uint8)   # one buffer, reused, as a render loop would
        for _ in range(2):
            scene.render(p_host, out=host_buf)
        t1 = time.perf_counter()
        for _ in range(n_solo):
            scene.render(p_host, out=host_buf)
        host_ms = (time.perf_counter() - t1) * 1e3 / n_solo
        # ... and rt_render_frames: F frames per call into host memory, batches that share their launches, the copy of one batch
        # (pinned staging + the host's memcpy into the caller's buffer) behind the render of the next
        host_frames_ms = {}
        for F in (4, 16, 64):
            host_many = np.zeros((F, H, W, 3), dtype=np.uint8)
            scene.render_frames(p_host, F, out=host_many)
            reps = max(2, 32 // F)
            t1 = time.perf_counter()
            for _ in range(reps):
                scene.render_frames(p_host, F, out=host_many)
            host_frames_ms[F] = (time.perf_counter() - t1) * 1e3 / (reps * F)
            del host_many
        # the same frames with --normals (the other flag set of config 2), B per launch like the headline
        p_n = params.copy()
        p_n.normals = 1
        for _ in range(2):
            scene.render_frames_device(p_n, B, sets[0].slices.data_ptr(), slice_rows=sets[0].max_rows, stream=streams[0].cuda_stream)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(4):
            scene.render_frames_device(p_n, B, sets[0].slices.data_ptr(), slice_rows=sets[0].max_rows, stream=streams[0].cuda_stream)
        torch.cuda.synchronize(dev)
        normals_ms = (time.perf_counter() - t1) * 1e3 / (4 * B)
        # the throughput variant (RT_SCHEDULE_WAVEFRONT_FP32: FP32 sphere test; NOT the parity mode, never the headline value): its
        # rate on the same frames and its mismatch count against the parity frame ("reported with its mismatch count")
        p_f = params.copy()
        p_f.schedule = rt.RT_SCHEDULE_WAVEFRONT_FP32
        parity = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        fast = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        scene.render_device(params, parity.data_ptr(), stream=streams[0].cuda_stream)
        scene.render_device(p_f, fast.data_ptr(), stream=streams[0].cuda_stream)
        for _ in range(2):
            scene.render_frames_device(p_f, B, sets[0].slices.data_ptr(), slice_rows=sets[0].max_rows, stream=streams[0].cuda_stream)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(4):
            scene.render_frames_device(p_f, B, sets[0].slices.data_ptr(), slice_rows=sets[0].max_rows, stream=streams[0].cuda_stream)
        torch.cuda.synchronize(dev)
        fp32_ms = (time.perf_counter() - t1) * 1e3 / (4 * B)
        diff = (parity.to(torch.int16) - fast.to(torch.int16)).abs().amax(dim=2)
        fp32_variant = {"what": "RT_SCHEDULE_WAVEFRONT_FP32: the sphere test of the traversal in FP32 (throughput variant, not the parity mode)",
                        "ms_per_frame": round(fp32_ms, 4), "Mrays_per_s": round(rays_frame / fp32_ms / 1e3, 1),
                        "pixels_off_by_more_than_1": int((diff > 1).sum().item()), "pixels_off_by_1": int((diff == 1).sum().item()),
                        "max_abs_diff": int(diff.max().item()), "pixels": W * H}

    if rank == 0:
        ms_per_step = elapsed * 1e3 / max(1, args.steps)
        if fp32_variant:
            fp32_variant["speedup_vs_parity"] = round(ms_per_step / fp32_variant["ms_per_frame"], 3)
        value = rays_frame * args.steps / elapsed / 1e6
        boxes_rank = int(c[1])
        # per launch of the dominant kernel = B frames of this rank
        alg_bytes = B * (nodes_rank * NODE_BYTES + prims_rank * SPHERE_BYTES_LAYOUT + rays_rank * RAY_STATE_BYTES)
        alg_bytes_survey = B * (nodes_rank * NODE_BYTES + prims_rank * SPHERE_BYTES_SURVEY + rays_rank * RAY_STATE_BYTES)
        alg_flops = B * (boxes_rank * FLOPS_PER_BOX_TEST + prims_rank * FLOPS_PER_SPHERE_TEST)
        kt = traverse_ms * 1e-3
        # What bounds k_traverse is VALU issue under divergence, not memory (the hierarchy is served by L2 / Infinity Cache:
        # measured HBM traffic is ~6 % of the algorithmic bytes).  The instruction and active-lane counts of a launch are fixed
        # by the build and the frame; they come from the committed rocprofv3 --pmc passes of this workload and are divided by
        # the launch duration measured live here.
        pmc = None
        tp = os.path.join(ROOT, "profiles", "pmc_c2.json")
        if os.path.exists(tp):
            try:
                t = json.load(open(tp))
                if t.get("spheres") == args.spheres and t.get("n_gpus", 1) == world and (args.width, args.height) == (1920, 1080):
                    if B == 1:
                        pmc = t
                    elif t.get("batch", {}).get("frames_per_launch") == B:   # counters of a launch of B frames
                        pmc = dict(t["batch"], build=t.get("build"), collected=t.get("collected"))
            except Exception:
                pmc = None
        roof = {"bound": "valu", "kernel": "k_traverse", "frames_per_launch": B, "kernel_ms": round(traverse_ms, 4), "kernel_launches": launches_alone,
                "kernel_ms_in_flight": round(traverse_ms_in_flight, 4), "kernel_launches_in_flight": launches,
                "kernel_ms_one_frame_per_launch": round(traverse_ms_one_frame, 4) if traverse_ms_one_frame else None,
                "unit": "Tlane-op/s", "peak": round(VALU_PEAK_TLANEOPS, 2), "achieved": None, "frac": None, "traffic": None,
                # SURVEY 8(d)'s own figure next to the utilisation: the ALGORITHMIC flops (22 per box test + 60 per sphere test) over
                # the kernel time against the FP32 vector peak.  `frac` counts every instruction the kernel issues as useful (sorting,
                # pushing and scheduling included); this one only the arithmetic the survey's formula prices.
                "frac_algorithmic": round(alg_flops / kt / 1e12 / VALU_PEAK_FP32_TFLOPS, 5) if kt > 0 else None,
                "algorithmic": {"flops_per_launch": alg_flops, "TFLOP_per_s": round(alg_flops / kt / 1e12, 3) if kt > 0 else None,
                                "frac_of_fp32_vector_peak": round(alg_flops / kt / 1e12 / VALU_PEAK_FP32_TFLOPS, 5) if kt > 0 else None,
                                "flops_per_ray": "22*box_tests + 60*sphere_tests (SURVEY 8(d))"},
                "node_bytes_per_s": {"GB_per_s_survey_constants": round(alg_bytes_survey / kt / 1e9, 1) if kt > 0 else None,
                                     "GB_per_s_this_layout": round(alg_bytes / kt / 1e9, 1) if kt > 0 else None,
                                     "bytes_per_launch_survey_constants": alg_bytes_survey, "bytes_per_launch_this_layout": alg_bytes,
                                     "bytes_per_ray": "64*node_fetches + B*sphere_tests + 16, B = 64 (SURVEY 8(d)) or 160 (DPrim 16 + FP64 DXform 144)",
                                     "note": "algorithmic bytes, served by L2 / Infinity Cache: not HBM traffic (see traffic)"}}
        if pmc and kt > 0:
            v = pmc["valu"]
            lane_ops = v["wave_instructions_per_launch"] * 64.0 * v["active_lane_fraction"]
            roof["achieved"] = round(lane_ops / kt / 1e12, 3)
            roof["frac"] = round(lane_ops / kt / 1e12 / VALU_PEAK_TLANEOPS, 4)
            roof["traffic"] = pmc.get("hbm_bytes_per_launch")
            roof["valu"] = {"wave_instructions_per_launch": v["wave_instructions_per_launch"], "active_lane_fraction": v["active_lane_fraction"],
                            "issue_fraction": round(v["wave_instructions_per_launch"] * 4.0 / (1024 * 2.4e9 * kt), 4)}
            roof["hbm"] = {"GB_per_s_measured": round(pmc["hbm_bytes_per_launch"] / kt / 1e9, 1) if pmc.get("hbm_bytes_per_launch") else None,
                           "frac_of_peak": round(pmc["hbm_bytes_per_launch"] / kt / 1e9 / HBM_PEAK_GBS, 4) if pmc.get("hbm_bytes_per_launch") else None,
                           "l2_hit_rate": pmc.get("l2_hit_rate")}
            if pmc.get("l2_request_bytes_per_launch"):
                # the memory level the node array lives in: requests the XCDs' L2s served (128 B each), against their 34.5 TB/s aggregate
                l2 = pmc["l2_request_bytes_per_launch"] / kt / 1e9
                roof["l2"] = {"GB_per_s": round(l2, 1), "peak_GB_per_s": L2_PEAK_GBS, "frac": round(l2 / L2_PEAK_GBS, 4),
                              "bytes_per_launch": pmc["l2_request_bytes_per_launch"], "hit_rate": pmc.get("l2_hit_rate")}
            roof["pmc_source"] = {"file": "profiles/pmc_c2.json", "build": pmc.get("build"), "collected": pmc.get("collected"),
                                  "note": "counter totals per launch come from that file (separate rocprofv3 --pmc passes); only the durations are measured in this run"}
        out = {
            "metric": "Mrays/s (primary+secondary) & ms/frame, 1080p Whitted, 1/2/4/8 GPUs", "value": round(value, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: %d random ellipsoids (default_rng(12345)), %dx%d, Whitted 1 spp, default flags, BVH on"
                                   % (args.spheres, W, H),
                       "partition": "interleaved row tiles of %d rows over %d GPU(s), u8 slices gathered to rank 0 (%s)"
                                    % (args.tile_rows, world, ("rt_gather_frames: RCCL send/recv + k_deinterleave" if native_exchange is not False else "torch.distributed gather over RCCL + index_select: the library's communicator did not come up") if args.backend == "nccl" else args.backend + " REHEARSAL, not a measurement"),
                       "rays_per_frame": rays_frame, "node_fetches_per_ray": round(nodes_frame / max(1, rays_frame), 2),
                       "box_tests_per_ray": round(boxes_frame / max(1, rays_frame), 2),
                       "prim_tests_per_ray": round(prims_frame / max(1, rays_frame), 2), "sampling": "seeded: a random position inside the pixel per ray as in main.cpp:288-302, frame f of a launch with seed + f" if args.sampling == "seeded" else "deterministic (RT_RNG_D0: pixel centres)",
                       "frames_per_launch": B, "slice_sets": S, "render_streams": n_streams,
                       "frames_per_exchange": B if world > 1 else None, "scene_create_ms": round(t_create * 1e3, 1),
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "rays_per_lane": os.environ.get("RT_RAYS_PER_LANE"),
                       "gathered_frame_equals_whole_frame": frame_ok},
            "ms_per_frame": round(ms_per_step, 4), "host_enqueue_ms_per_step": round(t_enqueued * 1e3 / max(1, args.steps), 4),
            "frame_latency_ms_in_flight": round(render_ms, 4),
            "phases": phases, "steady_state": steady,
            "ms_per_frame_solo": round(solo_ms, 4) if solo_ms else None,
            "Mrays_per_s_solo": round(rays_frame / solo_ms / 1e3, 1) if solo_ms else None,
            "ms_per_frame_host_buffer": round(host_ms, 4) if host_ms else None,
            "ms_per_frame_host_buffer_frames_per_call": {str(k): round(v, 4) for k, v in host_frames_ms.items()} or None,
            "ms_per_frame_normals_flag": round(normals_ms, 4) if normals_ms else None,
            "Mrays_per_s_normals_flag": round(rays_frame / normals_ms / 1e3, 1) if normals_ms else None,
            "fp32_variant": fp32_variant,
            "Mrays_per_s_host_buffer": round(rays_frame / host_ms / 1e3, 1) if host_ms else None,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            p1 = params.copy()
            p1.part_count, p1.part_index = 1, 0
            out["cpu_baseline"] = cpu_baseline(scene_desc, p1, args.cpu_row_step, args.cpu_passes)
        if world == 1 and not args.no_configs:
            scene.close()
            rt.release_workspaces()
            out["configs"] = other_configs(dev, with_cpu=not args.no_cpu_baseline)
        print(json.dumps(out), flush=True)
    scene.close()
    if world > 1:
        dist.barrier()
        rt.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
