#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json: "Mrays/sec at 1920x1080 depth-5; % HBM-roofline;
1/2/4/8-GPU scaling").

A *step* is one full frame: every rank traces its interleaved rows of the 1920x1080 image (fuel 5) with the HIP
kernels and, for N > 1, the tiles are gathered to rank 0 over RCCL and de-interleaved.  Frames are software-pipelined
(double-buffered tiles): the render of frame i overlaps the gather of frame i-1; all K renders and K gathers happen
inside the timed region.  Rays are *unique* rays
(SURVEY.md §8d): primary + shadow + reflection + refraction casts, counted by the kernel's counting variant in an
untimed pass (the count is deterministic).  Total work is fixed as N grows -> "strong" scaling.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config2|config3|config2_cones] [--no-cpu-baseline]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline.achieved` = algorithmic bytes of the dominant (only) kernel per launch
(SURVEY.md §8d formula on the kernel's own counters) / its average duration: HIP events recorded on the stream the
kernel is launched on around the K launches of the timed region, / K.  `cpu_baseline` = the CPU oracle (C++ restatement of the reference algorithm, NOT the Rust reference)
timed on a bounded pixel sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def make_workload(name):
    from raytracer_challenge_amd import scenes
    if name == "config2":
        cam, world = scenes.synthetic_analytic(n_primitives=512, seed=12345, cones=False, grouped=False, hsize=1920, vsize=1080)
        desc = "BASELINE configs[1]: synthetic analytic scene (3 planes + 512 spheres/cubes/cylinders, seed 12345), 1920x1080, fuel 5"
    elif name == "config2_cones":
        cam, world = scenes.synthetic_analytic(n_primitives=512, seed=12345, cones=True, grouped=True, hsize=1920, vsize=1080)
        desc = "configs[1] variant with 10% cones, grouped in 8 cells, 1920x1080, fuel 5"
    elif name == "config3":
        cam, world = scenes.chapter15_teapot("teapot_low.obj", 1920, 1080)
        desc = "BASELINE configs[2]: chapter15 teapot scene, teapot_low.obj (240 smooth triangles) + BVH, 1920x1080, fuel 5"
    elif name == "config3_high":
        cam, world = scenes.chapter15_teapot("teapot_high.obj", 1920, 1080)
        desc = "chapter15 teapot scene, teapot_high.obj (6320 smooth triangles) + BVH, 1920x1080, fuel 5"
    elif name == "config4":
        cam, world = scenes.chapter15_teapot("teapot_high.obj", 3840, 2160)
        desc = "BASELINE configs[3] on this many GPUs: chapter15 teapot scene, teapot_high.obj (6320 smooth triangles) + BVH, 3840x2160, fuel 8"
    elif name == "config5":
        import tempfile
        path = os.path.join(tempfile.gettempdir(), "rtc_heightfield_708x708_12345.obj")
        cam, world = scenes.synthetic_mesh(path)   # writes the OBJ (999 698 triangles, one group) if it is not there yet
        desc = "BASELINE configs[4] on this many GPUs: 999 698-triangle synthetic smooth mesh + Fractal/Simplex noise patterns, 3840x2160, fuel 8"
    else:
        raise SystemExit("unknown workload %r" % name)
    return cam, world, desc


def pmc_traffic(workload, path):
    """HBM bytes per launch (one-kernel path) / per frame (wavefront path) measured with rocprofv3 PMC passes for this
    workload on the device path this run uses (profiles/pmc_traffic.json), or None."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[workload]
        if ("wavefront" in t.get("path", "")) != (path == "wavefront"):
            return None, None
        return t["fetch_bytes"] + t["write_bytes"], t["source"]
    except Exception:
        return None, None


def capped_algorithmic_bytes(st, n_prims):
    """the counted figure, never more than 4x the ideal one-descent figure (guards against a bad BVH)."""
    import math
    from raytracer_challenge_amd.device import algorithmic_bytes
    ideal = 96 + 64 * math.ceil(math.log2(max(2, n_prims))) + 72 * 4
    return min(algorithmic_bytes(st), 4 * ideal * st["unique_rays"] + 24 * st["pixels"])


def cpu_threads():
    """Threads this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box hands one GPU a
    16-core share of a 256-thread host)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("RTC_CPU_THREADS", "16"))))


def cpu_baseline(world, cam, fuel, target_seconds=30.0):  # the short probe over-estimates the per-pixel cost ~2.5x: ~12 s measured
    """Oracle (CPU restatement) on a bounded, strided pixel sample of the same frame.  Returns the JSON object."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import oracle
    orc = oracle()
    nw = orc.build_world(world)
    total = cam.hsize * cam.vsize
    probe = np.arange(0, total, max(1, total // 2048), dtype=np.uint64)[:2048]
    threads = cpu_threads()
    _, _, st = orc.render_timed(nw, cam, fuel, probe, threads=threads)
    per_px = max(st.seconds / len(probe), 1e-9)
    n = int(min(total, max(4096, target_seconds / per_px)))
    stride = max(1, total // n)
    idx = np.arange(0, total, stride, dtype=np.uint64)
    _, _, st = orc.render_timed(nw, cam, fuel, idx, threads=threads)
    return {
        "value": st.unique_rays / st.seconds / 1e6, "unit": "Mrays/s", "cores": int(st.threads), "kind": "port",
        "sample": "every %d-th pixel of the frame (%d px, %.1f s): oracle = C++ restatement of the reference algorithm "
                  "(flat groups, all-hits + stable sort, per-light re-tracing), %d std::threads" % (stride, len(idx), st.seconds, threads),
        "reference_traced_mrays_per_s": st.traced_rays / st.seconds / 1e6,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--fuel", type=int, default=None, help="recursion depth (default: 5; 8 for config4/config5)")
    ap.add_argument("--inflight", type=int, default=0, help="frames in flight per GPU, each on its own scene copy and HIP stream "
                    "(default: 1 on one GPU, 3 on several, where the per-rank frames are small and latency-bound)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra-workloads", default="config3", help="comma list measured (untimed region) and reported under 'extra' at N=1")
    args = ap.parse_args()

    if args.fuel is None:
        args.fuel = 8 if args.workload in ("config4", "config5") else 5
    import torch
    import raytracer_challenge_amd as rt
    from raytracer_challenge_amd.device import DeviceRenderer, algorithmic_bytes

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world_size
    dist = None
    # RTC_BENCH_REHEARSE=1: rehearsal of the multi-rank code path on a one-GPU box (every rank on cuda:0, gloo gather through
    # host memory).  Its numbers mean nothing; the driver's N-GPU runs never set it.
    rehearse = world_size > 1 and os.environ.get("RTC_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    gather_dev = torch.device("cpu") if rehearse else dev

    hip = rt.hip_backend()  # raises if the HIP library is missing
    cam, world, desc = make_workload(args.workload)
    # F frames in flight: F copies of the uploaded scene, each with its own HIP stream (and wavefront buffers), frames dealt
    # round-robin.  A frame's per-level kernels are latency-bound at the small deep levels (and at the small per-rank frames of
    # an N-GPU run); the next frames' kernels fill the chip meanwhile.
    F = args.inflight if args.inflight > 0 else (1 if world_size == 1 else 3)
    nws = [hip.build_world(world) for _ in range(F)]
    drs = [DeviceRenderer(hip, w, cam, device=local_rank) for w in nws]
    nw, dr = nws[0], drs[0]
    H, V = cam.hsize, cam.vsize
    from raytracer_challenge_amd.parallel import FrameGatherer
    fg = FrameGatherer(H, V, rank, world_size, gather_dev, dist, n_buffers=F, tile_device=dev)

    def finish(i):
        """Frame i: wait for its render (marker 0 of its renderer), then gather its tiles to rank 0 (RCCL) and de-interleave."""
        drs[i % F].wait(0)
        if world_size > 1:
            fg.gather(i % F)
            torch.cuda.current_stream().synchronize()  # tiles[i % F] is free again once the gather has consumed it

    def run_frames(k):
        """k full frames, software-pipelined: up to F renders in flight; the gather of frame i - F runs behind them."""
        for i in range(k):
            if i >= F:
                finish(i - F)
            drs[i % F].render_rows_async(args.fuel, rank, world_size, fg.n_rows, fg.tiles[i % F])
            drs[i % F].record(0)
        for i in range(max(0, k - F), k):
            finish(i)

    # untimed: counting variant -> unique rays + algorithmic bytes of this rank's launch
    cst = dr.render_rows(args.fuel, rank, world_size, fg.n_rows, fg.tiles[0], count=True, sync=True)
    rays_local = torch.tensor([float(cst["unique_rays"])], dtype=torch.float64, device=gather_dev)
    if world_size > 1:
        dist.all_reduce(rays_local)
    rays_total = float(rays_local.item())

    # untimed: both device paths measured twice per renderer, the faster one kept
    paths = [d.tune(args.fuel, rank, world_size, fg.n_rows, fg.tiles[j]) for j, d in enumerate(drs)]
    path = paths[0]
    run_frames(args.warmup)
    for d in drs:
        d.check()
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize()
    for d in drs:
        d.sync()
        d.record(2)                   # stream markers 2..3 bracket the timed region's launches on each renderer's own stream
    t0 = time.perf_counter()
    run_frames(args.steps)
    for d in drs:
        d.record(3)
    for d in drs:
        d.sync()
    torch.cuda.synchronize()
    if world_size > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    for d in drs:
        d.check()                     # NaN / guard / overflow flags of the timed launches
    # device time per frame: throughput view (the timed region on the busiest stream / all frames) and latency view (a
    # stream's region / the frames it rendered: what rocprofv3's per-kernel durations add up to)
    used = [j for j in range(F) if j < args.steps]
    region_ms = [drs[j].elapsed_ms(2, 3) for j in used]
    frames_on = [len(range(j, args.steps, F)) for j in used]
    kernel_ms = [max(region_ms) / max(1, args.steps)]
    latency_ms = sum(r / f for r, f in zip(region_ms, frames_on)) / len(used)
    el = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
    if world_size > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    if rank == 0:
        # roofline per launch: a frame's own device time (latency view, comparable with rocprofv3's kernel durations); frames
        # in flight overlap, so the chip-wide rate is the throughput view
        throughput_ms = sum(kernel_ms) / len(kernel_ms)
        avg_kernel_ms = latency_ms
        alg_bytes = capped_algorithmic_bytes(cst, nw.primitive_count)
        achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "Mrays/s (unique rays: primary+shadow+reflection+refraction) at %dx%d depth-%d" % (H, V, args.fuel),
            "value": rays_total * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "hsize": H, "vsize": V, "fuel": args.fuel, "lights": nw.n_lights, "primitives": nw.primitive_count,
                       "partition": "rows interleaved by rank, RCCL gather to rank 0" if world_size > 1 else "single GPU",
                       "unique_rays_per_frame": rays_total, "rays_per_pixel": rays_total / (H * V)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(args.workload, path["path"])[0] if world_size == 1 else None, "traffic_source": pmc_traffic(args.workload, path["path"])[1],
                         "kernel": ("wavefront frame = wf_ts x%d + wf_shade x%d + wf_gather (dominant: wf_ts)" % (args.fuel + 2, args.fuel + 1))
                                   if path["path"] == "wavefront" else "rtc_trace_kernel",
                         "kernel_ms_avg": avg_kernel_ms, "frames_in_flight": F, "device_ms_per_frame_all_streams": throughput_ms,
                         "achieved_all_streams": alg_bytes / (throughput_ms * 1e-3) / 1e9, "path": path, "algorithmic_bytes_per_launch": alg_bytes,
                         "counters_rank0": {k: cst[k] for k in ("pixels", "unique_rays", "accel_nodes", "group_tests", "tri_tests", "analytic_tests", "rays_container")},
                         "note": "scene (%d B) is L2/Infinity-Cache resident; real HBM traffic ~ framebuffer only (SURVEY.md §8d)" % dr.info()["scene_device_bytes"]},
            "accelerator": dr.info(),
        }
        if world_size == 1:
            extra = {}
            for name in [w for w in args.extra_workloads.split(",") if w and w != args.workload]:
                c2, w2, d2 = make_workload(name)
                nw2 = hip.build_world(w2)
                dr2 = DeviceRenderer(hip, nw2, c2, device=local_rank)
                t2 = torch.zeros(c2.vsize * c2.hsize * 3, dtype=torch.float64, device=dev)
                s2 = dr2.render_rows(args.fuel, 0, 1, c2.vsize, t2, count=True)
                ms = [dr2.render_rows(args.fuel, 0, 1, c2.vsize, t2)["kernel_ms"] for _ in range(8)]  # the first four measure the paths
                ms = sum(ms[4:]) / len(ms[4:])
                extra[name] = {"workload": d2, "mrays_per_s_kernel": s2["unique_rays"] / ms / 1e3, "kernel_ms": ms, "unique_rays": s2["unique_rays"],
                               "roofline_achieved_GBs": capped_algorithmic_bytes(s2, nw2.primitive_count) / (ms * 1e-3) / 1e9, "accelerator": dr2.info(), "path": dr2.path_info()}
            out["extra"] = extra
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(world, cam, args.fuel)
        print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
