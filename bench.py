#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json: "Mrays/sec at 1920x1080 depth-5; % HBM-roofline;
1/2/4/8-GPU scaling").

A *step* is one full frame: every rank traces its interleaved rows of the 1920x1080 image (fuel 5) with the HIP
kernels and, for N > 1, the tiles are gathered to rank 0 over RCCL and de-interleaved.  Frames are software-pipelined:
F frames in flight per GPU (the SAME F at every N, default 3), each on its own copy of the uploaded scene and its own HIP
stream; all K renders and K gathers happen inside the timed region.  Rays are *unique* rays: primary +
shadow + reflection + refraction casts, counted by the kernels' counting variant in an untimed pass (the count is
deterministic).  Total work is fixed as N grows -> "strong" scaling.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config2|config3|...] [--inflight F] [--no-cpu-baseline] [--no-pmc]

`--gpus N` with N > 1 starts the N ranks itself (a child `python -m torch.distributed.run --nproc-per-node N bench.py ...`,
started before this process touches a GPU) unless it already runs under torch.distributed.run (WORLD_SIZE set), as the
driver launches it.  Rank 0 prints ONE JSON line:

  value / ms_per_step   whole-job unique rays of K frames / wall time of the timed region (barrier + synchronize on both sides)
  roofline              bound "hbm": algorithmic bytes per frame that go through the memory system (device.algorithmic_bytes:
                        SURVEY §8d units on the kernels' own counters; kernel-argument- and LDS-resident records count 0) / the frame's
                        device time, measured with HIP events on the scene's stream over sequential launches; `traffic` =
                        HBM bytes per frame from rocprofv3 PMC passes of THIS workload taken inside this run (child processes),
                        corrected as profiles/pmc_calibration.json says; `hbm_traffic_frac` = traffic / device time / peak
  valu                  what actually bounds the path: VALU pipe occupancy and SIMD lane utilisation from the same PMC passes
  parity                the frame the timed path wrote vs the CPU oracle on the cpu_baseline's pixel sample
  cpu_baseline          the CPU oracle (C++ restatement of the reference algorithm, NOT the Rust reference) on a bounded sample
  config3               the same measurements for BASELINE configs[2] (the teapot + BVH scene), at N = 1
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
N_SIMD = 256 * 4        # same guide: 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9        # same guide: max clock (the chip runs below it under load: busy fractions are lower bounds)


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
    elif name == "smoke":
        cam, world = scenes.synthetic_analytic(n_primitives=64, seed=12345, cones=False, grouped=False, hsize=64, vsize=36)
        desc = "rehearsal-size synthetic analytic scene, 64x36, fuel 5 (not a BASELINE config)"
    else:
        raise SystemExit("unknown workload %r" % name)
    return cam, world, desc


def default_fuel(workload):
    return 8 if workload in ("config4", "config5") else 5


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


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks as child processes (never exec: this process may already have touched a GPU)
# ---------------------------------------------------------------------------------------------------------------------
def self_launch(n):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for l in p.stdout.splitlines():
        if l.startswith("{") and '"metric"' in l:
            line = l
        else:
            print(l, file=sys.stderr)
    if line:
        print(line, flush=True)
    sys.exit(p.returncode if p.returncode else (0 if line else 1))


# ---------------------------------------------------------------------------------------------------------------------
# rocprofv3 PMC passes of this workload, taken by child processes of this run (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE
# do not fit one pass; PMC only ever with --kernel-trace)
# ---------------------------------------------------------------------------------------------------------------------
PMC_PASSES = [
    ["FETCH_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAVES", "SQ_BUSY_CYCLES"],
    ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"],
]


def pmc_calibration():
    """Bytes per counter unit for FETCH_SIZE / WRITE_SIZE in this path's access patterns (scripts/pmc_calibrate.hip measured on
    known byte counts, as the guide prescribes for access widths other than 16 B per lane)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_calibration.json")))
    except Exception:
        return {"fetch_bytes_per_unit": 1024.0, "write_bytes_per_unit": 1024.0, "source": "uncalibrated: rocprofv3 reports KiB"}


def frame_kernels(path, fuel):
    """(kernel name prefix, launches per frame) of the timed (non-counting) kernels of a device path."""
    if "wavefront" in path:
        return [("wf_ts<false", fuel + 2), ("wf_shade<false", fuel + 1), ("wf_gather", 1)]
    return [("rtc_trace_kernel<false", 1)]


def pmc_measure(workload, fuel, path, timeout_s=240):
    """Runs the PMC passes; returns {counter: value per FRAME of the chosen device path} (mean per dispatch x dispatches per
    frame, per kernel, summed) plus per-kernel tables, or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    from collections import defaultdict
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    kernels = frame_kernels(path, fuel)
    per_frame, per_kernel = defaultdict(float), {}
    for counters in PMC_PASSES:
        out = tempfile.mkdtemp(prefix="rtc_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
               "--pmc-child", "--workload", workload, "--fuel", str(fuel), "--steps", "3", "--warmup", "1", "--inflight", "1"]
        # the child renders with the parent's device path (no tuning launches of its own under the profiler)
        env = dict(os.environ, TMPDIR="/tmp", RTC_KERNEL="4" if "wavefront" in path else "1")
        try:
            p = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            shutil.rmtree(out, ignore_errors=True)
            return None, "rocprofv3 pass timed out"
        if p.returncode != 0:
            shutil.rmtree(out, ignore_errors=True)
            return None, "rocprofv3 pass failed (rc %d): %s" % (p.returncode, p.stdout[-300:])
        acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                kn = row.get("Kernel_Name", "").replace("void ", "")
                for prefix, _ in kernels:
                    if kn.startswith(prefix):
                        c = acc[prefix][row["Counter_Name"]]
                        c[0] += 1
                        c[1] += float(row["Counter_Value"])
        shutil.rmtree(out, ignore_errors=True)
        for prefix, per in kernels:
            for cn, (n, tot) in acc[prefix].items():
                per_frame[cn] += tot / n * per
                per_kernel.setdefault(prefix, {})[cn] = {"mean_per_dispatch": tot / n, "dispatches_profiled": n, "dispatches_per_frame": per}
        missing = [c for c in counters if c not in per_frame]
        if missing:
            return None, "counters missing from the rocprofv3 output: %s" % missing
    return {"per_frame": dict(per_frame), "per_kernel": per_kernel}, None


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline + parity sample (the oracle: test infrastructure, used here only as the checker / the baseline)
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(world, cam, fuel, target_seconds=30.0):  # the short probe over-estimates the per-pixel cost ~2.5x: ~12 s measured
    """Oracle (CPU restatement) on a bounded, strided pixel sample of the same frame.  Returns (json object, idx, rgb, hits)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import oracle
    orc = oracle()
    nw = orc.build_world(world)
    total = cam.hsize * cam.vsize
    threads = cpu_threads()
    # probe: a strided sample grown until it has cost >= 1.5 s (the oracle's flat lists make a pixel of the 10^6-triangle scene cost
    # seconds, a pixel of the analytic scene milliseconds), then the sample that fits target_seconds
    n_probe, per_px = 64, None
    while True:
        probe = np.arange(0, total, max(1, total // n_probe), dtype=np.uint64)[:n_probe]
        _, _, st = orc.render_timed(nw, cam, fuel, probe, threads=threads)
        per_px = max(st.seconds / len(probe), 1e-9)
        if st.seconds >= 1.5 or n_probe >= 4096:
            break
        n_probe *= 4
    n = int(min(total, max(64, target_seconds / per_px)))
    stride = max(1, total // n)
    idx = np.arange(0, total, stride, dtype=np.uint64)
    rgb, hits, st = orc.render_timed(nw, cam, fuel, idx, threads=threads)
    obj = {
        "value": st.unique_rays / st.seconds / 1e6, "unit": "Mrays/s", "cores": int(st.threads), "kind": "port",
        "sample": "every %d-th pixel of the frame (%d px, %.1f s): oracle = C++ restatement of the reference algorithm "
                  "(flat groups, all-hits + stable sort, per-light re-tracing), %d std::threads" % (stride, len(idx), st.seconds, threads),
        "reference_traced_mrays_per_s": st.traced_rays / st.seconds / 1e6,
    }
    return obj, idx, rgb, hits


def parity_block(idx, ref_rgb, ref_hits, gpu_rgb_full, gpu_hits_full, timed_frame_equal):
    """SURVEY §8(d) 'parity check accompanying every number': the oracle's sample against the same pixels of the GPU frame."""
    import numpy as np
    ii = idx.astype(np.int64)
    g_rgb, g_hits = gpu_rgb_full[ii], gpu_hits_full[ii]
    bad = (g_hits["prim"] != ref_hits["prim"]) | (g_hits["push_idx"] != ref_hits["push_idx"]) | (g_hits["t"].view(np.uint64) != ref_hits["t"].view(np.uint64))
    return {"pixels": int(len(idx)), "max_abs_drgb": float(np.abs(g_rgb - ref_rgb).max()) if len(idx) else 0.0, "tolerance": 1e-5,
            "primary_hit_mismatches": int(bad.sum()), "host_fallback_pixels": 0,
            "timed_frame_equals_checked_frame": bool(timed_frame_equal),
            "checked": "full frame rendered through rtc_render on the device path of the timed region, compared with the oracle on the cpu_baseline sample"}


def host_pixel_times(hip, dr, nw, cam, fuel, reps=5):
    """ms per blocking host-pixel render of the whole frame (the drop-in call), median of `reps`."""
    import ctypes as C
    import numpy as np
    from raytracer_challenge_amd.device import RtcCameraC, RtcStatsC
    lib = hip.lib
    lib.rtc_render.restype = C.c_int
    lib.rtc_render.argtypes = [C.c_void_p, C.POINTER(RtcCameraC), C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rtc_render_rgb8.restype = C.c_int
    lib.rtc_render_rgb8.argtypes = [C.c_void_p, C.POINTER(RtcCameraC), C.c_int32, C.c_void_p, C.c_void_p]
    n = cam.hsize * cam.vsize
    from raytracer_challenge_amd.backend import HIT_DTYPE

    def timed(fn):
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t) * 1e3)
        return float(np.median(ts))

    def call(rgb, hits):
        rc = lib.rtc_render(dr.scene, C.byref(dr.cam), fuel, None, 0, n, rgb.ctypes.data, hits.ctypes.data if hits is not None else None, None)
        if rc != 0:
            raise SystemExit("rtc_render: %s" % lib.rtc_last_error())

    def call8(buf):
        rc = lib.rtc_render_rgb8(dr.scene, C.byref(dr.cam), fuel, buf.ctypes.data, None)
        if rc != 0:
            raise SystemExit("rtc_render_rgb8: %s" % lib.rtc_last_error())

    keep_rgb, keep_hits, keep8 = np.zeros((n, 3)), np.zeros(n, dtype=HIT_DTYPE), np.zeros(3 * n, dtype=np.uint8)
    call(keep_rgb, keep_hits)
    call8(keep8)

    def timed_fresh(make, fn):
        """`reps` calls, each into a destination allocated BEFORE the clock starts and never touched (numpy.empty: no pages yet)."""
        bufs = [make() for _ in range(reps)]
        ts = []
        for b in bufs:
            t = time.perf_counter()
            fn(b)
            ts.append((time.perf_counter() - t) * 1e3)
        return float(np.median(ts))

    out = {
        "f64_reused_buffer_ms": timed(lambda: call(keep_rgb, None)),
        "f64_fresh_buffer_ms": timed_fresh(lambda: np.empty((n, 3)), lambda b: call(b, None)),
        "f64_with_hits_reused_buffers_ms": timed(lambda: call(keep_rgb, keep_hits)),
        "rgb8_reused_buffer_ms": timed(lambda: call8(keep8)),
        "rgb8_fresh_buffer_ms": timed_fresh(lambda: np.empty(3 * n, dtype=np.uint8), call8),
        "bytes": {"f64": 24 * n, "hits": 16 * n, "rgb8": 3 * n},
        "note": "wall clock of the blocking C-ABI call (kernels + one copy per array into the caller's pageable memory), median of %d.  reused = the "
                "destination's pages exist (PCIe rate: ~1.0 ms per 50 MB); fresh = a never-touched numpy.empty destination per call: the copy also "
                "faults the caller's new pages in (kernel mm work of the caller's allocation, ~+1.4 ms per 50 MB in scripts/d2h_probe.hip)" % reps,
    }
    return out


# ---------------------------------------------------------------------------------------------------------------------
# The reference's own call pattern: one par_render per process (src/bin/*.rs: build the world, render once at 4096x2160, exit).
# Measured in a fresh child process (cold code objects, no warm allocator): OBJ parse + flatten + accelerator build + upload,
# the first render's kernels, the copy to host memory.
# ---------------------------------------------------------------------------------------------------------------------
ONE_SHOT = {"config3_4096x2160": ("config3", 4096, 2160), "config2": ("config2", 1920, 1080)}


def one_shot_child(name):
    import ctypes as C
    import numpy as np
    t_start = time.perf_counter()
    import raytracer_challenge_amd as rt
    from raytracer_challenge_amd import scenes
    from raytracer_challenge_amd.device import DeviceRenderer, RtcCameraC, RtcStatsC
    workload, H, V = ONE_SHOT[name]
    if workload == "config3":
        cam, world = scenes.chapter15_teapot("teapot_low.obj", H, V)
    else:
        cam, world, _ = make_workload(workload)
    fuel = default_fuel(workload)
    hip = rt.hip_backend()
    t0 = time.perf_counter()
    nw = hip.build_world(world)              # host mirror: Shape::shape, composite, parse_obj (the OBJ file is read and parsed here)
    t1 = time.perf_counter()
    dr = DeviceRenderer(hip, nw, cam)        # flatten + accelerator build + upload (first HIP call of the process: context creation)
    t2 = time.perf_counter()
    lib = hip.lib
    lib.rtc_render.restype = C.c_int
    lib.rtc_render.argtypes = [C.c_void_p, C.POINTER(RtcCameraC), C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(RtcStatsC)]
    n = H * V
    rgb = np.empty((n, 3))
    st = RtcStatsC()
    t3 = time.perf_counter()
    rc = lib.rtc_render(dr.scene, C.byref(dr.cam), fuel, None, 0, n, rgb.ctypes.data, None, None)
    t4 = time.perf_counter()
    if rc != 0:
        raise SystemExit("rtc_render: %s" % lib.rtc_last_error())
    pi = dr.path_info()
    # the same call again (warm): what the first call paid on top (code object load, scratch, queues, page faults)
    rgb2 = np.empty((n, 3))
    t5 = time.perf_counter()
    lib.rtc_render(dr.scene, C.byref(dr.cam), fuel, None, 0, n, rgb2.ctypes.data, None, None)
    t6 = time.perf_counter()
    lib.rtc_render(dr.scene, C.byref(dr.cam), fuel, None, 0, n, rgb2.ctypes.data, None, C.byref(st))   # counting variant: device time of the kernels
    obj_parse_ms = None
    if workload == "config3":
        import ctypes
        lib.rtw_parse_obj.restype = ctypes.c_void_p
        tp = time.perf_counter()
        e = hip._element(world.elements[-1], {}, []) if world.elements[-1].tag == "obj" else None
        obj_parse_ms = (time.perf_counter() - tp) * 1e3 if e else None
    print(json.dumps({"one_shot": name, "hsize": H, "vsize": V, "fuel": fuel,
                      "import_ms": (t0 - t_start) * 1e3, "build_world_ms": (t1 - t0) * 1e3, "create_ms": (t2 - t1) * 1e3,
                      "first_render_to_host_ms": (t4 - t3) * 1e3, "second_render_to_host_ms": (t6 - t5) * 1e3,
                      "kernel_ms_of_a_counting_launch_not_the_timed_kernels": st.kernel_ms, "obj_parse_ms": obj_parse_ms,
                      "total_ms": (t4 - t0) * 1e3, "first_launch_path": "wavefront" if st.n_launches > 1 else "one kernel", "path_info": pi,
                      "equal": bool(np.array_equal(rgb, rgb2))}), flush=True)


def one_shot_measure(timeout_s=300):
    out = {}
    for name in ONE_SHOT:
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one-shot-child", name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            out[name] = json.loads(line[-1]) if line else {"error": (p.stderr or "")[-300:]}
        except Exception as e:  # a failed side measurement must not lose the headline line
            out[name] = {"error": repr(e)}
    out["note"] = ("fresh child process per entry: build_world = host mirror incl. OBJ parse; create = flatten + accelerator + upload (+ HIP context); "
                   "first_render_to_host = one blocking rtc_render of the whole frame into a new host buffer; total = build_world + create + first render")
    return out


class Runtime:
    """Where tiles live and how the device is synchronised.  The product runtime is the GPU (librtc_amd.so, HIP streams, RCCL).
    RTC_BENCH_CPU_STANDIN=1 (tests/test_bench_cpu.py only) swaps in the CPU emulator of the kernel source (tests/cpu_emu, test
    infrastructure) so that the launch / partition / gather / reporting code of this file can run without a GPU: such a run is a
    rehearsal of the plumbing, never a measurement, and says so in its JSON line."""

    def __init__(self, torch, local_rank, standin):
        self.torch, self.local_rank, self.standin = torch, local_rank, standin
        if standin:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from emu_lib import emu
            self.backend = emu()
            self.dev = torch.device("cpu")
        else:
            import raytracer_challenge_amd as rt
            torch.cuda.set_device(local_rank)
            self.backend = rt.hip_backend()  # raises if the HIP library is missing
            self.dev = torch.device("cuda", local_rank)

    def synchronize(self):
        if not self.standin:
            self.torch.cuda.synchronize()

    def renderer(self, world, cam):
        from raytracer_challenge_amd.device import DeviceRenderer
        return DeviceRenderer(self.backend, world, cam, device=self.local_rank, _cpu_standin=self.standin)


# ---------------------------------------------------------------------------------------------------------------------
def measure(args, rtm, dist, workload, fuel, steps, warmup, F, rank, world_size, rehearse, want_pmc, want_cpu, partition="bands"):
    """One workload on this process group.  Returns the dict of measurements (rank 0: complete; other ranks: partial)."""
    import numpy as np
    from raytracer_challenge_amd.device import algorithmic_bytes
    from raytracer_challenge_amd.parallel import FrameGatherer, FrameRoundGatherer
    torch, hip, dev = rtm.torch, rtm.backend, rtm.dev
    gather_dev = torch.device("cpu") if rehearse else dev
    cam, world, desc = make_workload(workload)
    t_b0 = time.perf_counter()
    nws = [hip.build_world(world)]             # host mirror of the scene: Shape::shape, composite, ObjParser::parse_obj (reads + parses the OBJ file)
    t_b1 = time.perf_counter()
    drs = [rtm.renderer(nws[0], cam)]          # flatten + accelerator build + upload
    t_b2 = time.perf_counter()
    nws += [hip.build_world(world) for _ in range(F - 1)]
    drs += [rtm.renderer(w, cam) for w in nws[1:]]
    nw, dr = nws[0], drs[0]
    ingest = {"build_world_ms_incl_obj_parse": (t_b1 - t_b0) * 1e3, "scene_create_ms_flatten_accelerator_upload": (t_b2 - t_b1) * 1e3,
              "note": "first of the F scene copies of this process (the very first create of a process also pays HIP context creation)"}
    H, V = cam.hsize, cam.vsize
    # How N GPUs share the K frames of the timed region:
    #   bands   every frame is cut into 8-row bands dealt round-robin over the ranks, one gather per frame (the north-star partition:
    #           a frame's latency drops, but a ~2 ms frame's per-rank share is latency-bound: scripts/partition_probe.py predicts 3.8x at N = 8);
    #   frames  whole frames round-robin over the ranks, a round of N frames delivered to rank 0 with one gather (same bytes per frame).
    frames_mode = partition == "frames" and world_size > 1
    if frames_mode:
        fg = FrameRoundGatherer(H, V, rank, world_size, gather_dev, dist, n_buffers=F, tile_device=dev)
        part, n_parts = 0, 1
    else:
        fg = FrameGatherer(H, V, rank, world_size, gather_dev, dist, n_buffers=F, tile_device=dev)
        part, n_parts = rank, world_size

    def finish(i):
        """Step i: wait for its render (marker 0 of its renderer), then gather to rank 0 (RCCL) (and de-interleave)."""
        drs[i % F].wait(0)
        if world_size > 1:
            fg.gather(i % F)
            if not rtm.standin:
                torch.cuda.current_stream().synchronize()  # tiles[i % F] is free again once the gather has consumed it

    def run_frames(k):
        """k full frames, software-pipelined: up to F renders in flight per GPU; the gather of step i - F runs behind them.
        bands: a step is a frame (every rank renders its bands); frames: a step is a round of up to N frames, one per rank."""
        n_steps = (k + world_size - 1) // world_size if frames_mode else k
        for i in range(n_steps):
            if i >= F:
                finish(i - F)
            if not frames_mode or i * world_size + rank < k:   # (the last round of K frames may be short)
                drs[i % F].render_rows_async(fuel, part, n_parts, fg.n_rows, fg.tiles[i % F], band_rows=fg.band_rows)
            drs[i % F].record(0)
        for i in range(max(0, n_steps - F), n_steps):
            finish(i)

    def barrier():
        if world_size > 1:
            dist.barrier()
        rtm.synchronize()
        for d in drs:
            d.sync()

    # untimed: counting variant -> unique rays + work counters of this rank's launch
    cst = dr.render_rows(fuel, part, n_parts, fg.n_rows, fg.tiles[0], count=True, sync=True, band_rows=fg.band_rows)
    rays_local = torch.tensor([float(cst["unique_rays"])], dtype=torch.float64, device=gather_dev)
    if world_size > 1 and not frames_mode:
        dist.all_reduce(rays_local)
    rays_total = float(rays_local.item())   # unique rays of ONE whole frame

    # untimed: both device paths measured twice per renderer, the faster one kept
    paths = [d.tune(fuel, part, n_parts, fg.n_rows, fg.tiles[j], band_rows=fg.band_rows) for j, d in enumerate(drs)]
    path = paths[0]
    run_frames(warmup)
    for d in drs:
        d.check()
    barrier()
    t0 = time.perf_counter()
    run_frames(steps)
    barrier()
    elapsed = time.perf_counter() - t0
    for d in drs:
        d.check()                     # error state of EVERY launch of the timed region (sticky until read)
    el = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
    if world_size > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    # untimed: the frame's own device time, one frame at a time (what rocprofv3's per-kernel durations add up to): HIP events on
    # the scene's stream around n_seq sequential launches
    n_seq = max(5, min(50, steps))
    dr.record(2)
    for _ in range(n_seq):
        dr.render_rows_async(fuel, part, n_parts, fg.n_rows, fg.tiles[0], band_rows=fg.band_rows)
    dr.record(3)
    dr.sync()
    dr.check()
    seq_ms = dr.elapsed_ms(2, 3) / n_seq
    timed_tile = fg.tiles[0].clone()

    res = {"workload": workload, "desc": desc, "H": H, "V": V, "fuel": fuel, "steps": steps, "warmup": warmup, "elapsed": elapsed, "rays_total": rays_total,
           "seq_ms": seq_ms, "path": path, "counters": cst, "n_lights": nw.n_lights, "primitives": nw.primitive_count, "info": dr.info(), "F": F,
           "partition": "frames" if frames_mode else "bands", "ingest": ingest}
    if rank != 0 or args.pmc_child:
        return res
    alg = algorithmic_bytes(cst, path["path"], nw.primitive_count, lds_tables=dr.info().get("wavefront_lds_bytes_per_block", 0) > 0)
    achieved = alg["memory"] / (seq_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "kernel": ("wavefront frame = wf_ts x%d + wf_shade x%d + wf_gather (dominant: wf_ts)" % (fuel + 2, fuel + 1)) if "wavefront" in path["path"] else "rtc_trace_kernel",
            "kernel_ms_avg": seq_ms, "kernel_ms_measured_over": "%d sequential launches after the timed region, HIP events on the scene's stream" % n_seq,
            "algorithmic_bytes_per_launch": alg["memory"], "algorithmic_bytes": alg, "path": path,
            "counters_rank0": {k: cst[k] for k in ("pixels", "unique_rays", "rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "rays_container", "accel_nodes",
                                                   "accel_nodes_kernarg", "group_tests", "tri_tests", "analytic_tests", "analytic_tests_kernarg", "light_grid_cells")},
            "note": "not the binding limit: the scene (%d B) is L2 / Infinity-Cache resident and most algorithmic bytes are served on chip; "
                    "hbm_traffic_frac is the real HBM share, `valu` the pipe that binds" % dr.info()["scene_device_bytes"]}
    res["roofline"] = roof
    if world_size == 1:
        if want_pmc:
            pm, why = pmc_measure(workload, fuel, path["path"])
            if pm is None:
                roof["traffic_note"] = why
            else:
                cal = pmc_calibration()
                pf = pm["per_frame"]
                fetch, write = pf["FETCH_SIZE"] * cal["fetch_bytes_per_unit"], pf["WRITE_SIZE"] * cal["write_bytes_per_unit"]
                roof["traffic"] = fetch + write
                roof["traffic_detail"] = {"fetch_bytes": fetch, "write_bytes": write, "calibration": {"fetch_bytes_per_unit": cal["fetch_bytes_per_unit"], "write_bytes_per_unit": cal["write_bytes_per_unit"], "source": "profiles/pmc_calibration.json (scripts/pmc_calibrate.sh: measured on known byte counts)"}, "l2_hit_rate": pf["TCC_HIT_sum"] / max(1.0, pf["TCC_HIT_sum"] + pf["TCC_MISS_sum"]),
                                          "source": "rocprofv3 --kernel-trace --pmc, two passes run by this bench invocation; per frame = mean per dispatch x dispatches per frame, per kernel"}
                roof["hbm_traffic_frac"] = (fetch + write) / (seq_ms * 1e-3) / (HBM_PEAK_GBS * 1e9)
                busy_cycles = 4.0 * pf["SQ_ACTIVE_INST_VALU"]   # the SQ counts quad-cycles
                res["valu"] = {
                    "bound": "valu issue (f64 vector pipe) x SIMD lane utilisation",
                    "insts_valu_per_frame": pf["SQ_INSTS_VALU"],
                    "valu_busy_frac": busy_cycles / (seq_ms * 1e-3 * CLOCK_HZ * N_SIMD),
                    "lane_utilisation": pf["SQ_THREAD_CYCLES_VALU"] / (64.0 * max(1.0, pf["SQ_ACTIVE_INST_VALU"])),
                    "wait_frac_of_wave_cycles": pf["SQ_WAIT_ANY"] / max(1.0, pf["SQ_WAVE_CYCLES"]),
                    "cycles_per_valu_inst": busy_cycles / max(1.0, pf["SQ_INSTS_VALU"]),
                    "formulae": "valu_busy_frac = 4 x SQ_ACTIVE_INST_VALU / (kernel s x 2.4 GHz x 1024 SIMDs); lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); "
                                "profiled clocks are lower than 2.4 GHz, so the busy fraction is a lower bound",
                    "per_kernel": pm["per_kernel"],
                }
    else:
        # N ranks: the frame rank 0 gathered (every rank's interleaved rows, de-interleaved) against the same frame rendered by rank 0's
        # device alone through rtc_render — the frame the N = 1 line checks against the oracle
        single, _ = hip.render(nw, cam, fuel, want_hits=False)
        if frames_mode:   # every rank's last delivered frame against rank 0's own render of the frame
            got = fg.frames.detach().cpu().numpy().reshape(world_size, -1, 3)
            same = all(bool(np.array_equal(got[r], single)) for r in range(world_size))
            worst = float(max(np.abs(got[r] - single).max() for r in range(world_size)))
        else:
            gathered = fg.image.detach().cpu().numpy().reshape(-1, 3)
            same, worst = bool(np.array_equal(gathered, single)), float(np.abs(gathered - single).max())
        res["parity"] = {"pixels": int(H * V), "gathered_frame_equals_single_gpu_frame": same,
                         "max_abs_drgb_vs_single_gpu_frame": worst,
                         "checked": "the frame(s) rank 0 gathered in the timed path (bands: de-interleaved; frames: the last frame every rank delivered) against "
                                    "rtc_render of the whole frame on rank 0's device; the oracle comparison of that frame is the parity block of the N = 1 line"}
    if world_size == 1:
        # PCIe-inclusive figures: Image::par_render returns host pixels (src/image.rs:76-80).  rtc_render / rtc_render_rgb8 into
        # a FRESH destination every call (what `-> Image` means: a new Vec whose pages do not exist yet), into a reused one, and
        # with the primary-hit channel; wall clock around the blocking call.
        res["host_pixels"] = host_pixel_times(hip, dr, nw, cam, fuel)
        res["ms_per_step_incl_d2h"] = res["host_pixels"]["f64_reused_buffer_ms"]
        rgb_full, hits_full = hip.render(nw, cam, fuel)
        if want_cpu:
            base, idx, ref_rgb, ref_hits = cpu_baseline(world, cam, fuel)
            res["cpu_baseline"] = base
            same = bool(np.array_equal(timed_tile.cpu().numpy()[: H * V * 3].reshape(-1, 3), rgb_full))
            res["parity"] = parity_block(idx, ref_rgb, ref_hits, rgb_full, hits_full, same)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed frames (default 300: >= 0.5 s of device time at ~2 ms/frame)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--fuel", type=int, default=None, help="recursion depth (default: 5; 8 for config4/config5)")
    ap.add_argument("--inflight", type=int, default=3, help="frames in flight per GPU, each on its own scene copy and HIP stream; the same at every N")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC child passes (roofline.traffic = null)")
    ap.add_argument("--extra-workloads", default="config3", help="comma list of further workloads measured in full after the headline one, at N=1")
    ap.add_argument("--partition", default="auto", choices=["auto", "bands", "frames"],
                    help="N > 1: bands = every frame cut into 8-row bands over the ranks (one gather per frame); frames = whole frames round-robin "
                         "over the ranks (one gather per round of N frames); auto = frames when at least 2 N frames are timed, else bands")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--one-shot-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-one-shot", action="store_true", help="skip the one-shot (fresh process, one render) measurements")
    args = ap.parse_args()
    if args.fuel is None:
        args.fuel = default_fuel(args.workload)

    if args.one_shot_child:
        one_shot_child(args.one_shot_child)
        return
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)          # does not return
    args.gpus = world_size

    import torch
    standin = os.environ.get("RTC_BENCH_CPU_STANDIN") == "1"
    dist = None
    # RTC_BENCH_REHEARSE=1: rehearsal of the multi-rank code path on a one-GPU box (every rank on cuda:0, gloo gather through
    # host memory).  Its numbers mean nothing; the driver's N-GPU runs never set it.
    rehearse = world_size > 1 and (standin or os.environ.get("RTC_BENCH_REHEARSE") == "1")
    if rehearse:
        local_rank = 0
    backend_name = None
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not standin:
            torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        backend_name = dist.get_backend()
    rtm = Runtime(torch, local_rank, standin)
    if standin:
        args.no_pmc = True

    if args.pmc_child:  # profiled by the parent's rocprofv3: the frames of the workload, nothing else
        measure(args, rtm, dist, args.workload, args.fuel, args.steps, args.warmup, 1, 0, 1, False, False, False)
        return

    F = max(1, args.inflight)
    partition = args.partition
    if partition == "auto":
        partition = "frames" if world_size > 1 and args.steps >= 2 * world_size else "bands"
    if partition == "frames" and args.steps < world_size:
        partition = "bands"   # fewer frames than GPUs: nothing to deal out
    m = measure(args, rtm, dist, args.workload, args.fuel, args.steps, args.warmup, F, rank, world_size, rehearse,
                want_pmc=not args.no_pmc, want_cpu=not args.no_cpu_baseline, partition=partition)
    band = None
    if world_size > 1 and m["partition"] == "frames":
        # the north-star partition beside it: every frame cut into bands over the ranks (what a single frame's latency gets from N GPUs)
        band = measure(args, rtm, dist, args.workload, args.fuel, max(2, min(args.steps, 60)), min(args.warmup, 3), F, rank, world_size, rehearse,
                       want_pmc=False, want_cpu=False, partition="bands")
    if rank == 0:
        H, V = m["H"], m["V"]
        out = {
            "metric": "Mrays/s (unique rays: primary+shadow+reflection+refraction) at %dx%d depth-%d" % (H, V, m["fuel"]),
            "value": m["rays_total"] * m["steps"] / m["elapsed"] / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world_size, "steps": m["steps"], "warmup": m["warmup"],
            "ms_per_step": m["elapsed"] / m["steps"] * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if not standin else "synthetic; CPU STAND-IN REHEARSAL of the bench plumbing (kernel source emulated on the CPU): not a measurement",
            "config": {"workload": m["desc"], "hsize": H, "vsize": V, "fuel": m["fuel"], "lights": m["n_lights"], "primitives": m["primitives"],
                       "partition": ("single GPU" if world_size == 1 else
                                     "frames: whole frames round-robin over the ranks, F in flight per GPU, a round of N frames gathered to rank 0 over RCCL (same bytes per frame as bands)"
                                     if m["partition"] == "frames" else "bands: every frame in 8-row bands dealt round-robin by rank, RCCL gather to rank 0"),
                       "frames_in_flight": F, "process_group": {"world_size": world_size, "backend": backend_name},
                       "unique_rays_per_frame": m["rays_total"], "rays_per_pixel": m["rays_total"] / (H * V),
                       "ray_count": "rays of the reference's de-duplicated ray tree (device counters = the oracle's); on the one-kernel path a shadow ray "
                                    "towards a light behind the surface is counted and answered without a traversal (DESIGN.md 4.13)"},
            "roofline": m["roofline"],
            "accelerator": dict(m["info"], ingest=m["ingest"]),
        }
        for k in ("valu", "parity", "cpu_baseline", "ms_per_step_incl_d2h", "host_pixels"):
            if k in m:
                out[k] = m[k]
        if band is not None:
            out["band_partition"] = {"value": band["rays_total"] * band["steps"] / band["elapsed"] / 1e6, "unit": "Mrays/s", "steps": band["steps"],
                                     "ms_per_step": band["elapsed"] / band["steps"] * 1e3, "parity": band.get("parity"),
                                     "note": "the same job with every frame cut into 8-row bands over the ranks (one gather per frame): what ONE frame's latency gets "
                                             "from N GPUs; its per-rank share is latency-bound (DESIGN.md section 6)"}
        if world_size == 1 and not args.no_one_shot and not standin:
            out["one_shot"] = one_shot_measure()
        if world_size == 1:
            for name in [w for w in args.extra_workloads.split(",") if w and w != args.workload]:
                fuel2 = default_fuel(name)
                e = measure(args, rtm, dist, name, fuel2, args.steps, args.warmup, F, 0, 1, False,
                            want_pmc=not args.no_pmc, want_cpu=not args.no_cpu_baseline)
                ent = {"workload": e["desc"], "value": e["rays_total"] * e["steps"] / e["elapsed"] / 1e6, "unit": "Mrays/s", "steps": e["steps"],
                       "ms_per_step": e["elapsed"] / e["steps"] * 1e3, "unique_rays_per_frame": e["rays_total"], "roofline": e["roofline"], "accelerator": dict(e["info"], ingest=e["ingest"])}
                for k in ("valu", "parity", "cpu_baseline", "ms_per_step_incl_d2h", "host_pixels"):
                    if k in e:
                        ent[k] = e[k]
                out[name] = ent
                # the headline numbers of every further workload inside the object the driver parses
                out["config"].setdefault("secondary", {})[name] = {
                    "workload": e["desc"], "value": ent["value"], "unit": "Mrays/s", "ms_per_step": ent["ms_per_step"],
                    "kernel_ms_sequential": e["seq_ms"], "path": e["path"]["path"], "roofline_frac": e["roofline"]["frac"],
                    "roofline_achieved_gbs": e["roofline"]["achieved"], "ms_per_step_incl_d2h": e.get("ms_per_step_incl_d2h"),
                    "parity_max_abs_drgb": (e.get("parity") or {}).get("max_abs_drgb"), "primary_hit_mismatches": (e.get("parity") or {}).get("primary_hit_mismatches")}
        # key order of the one line: the bulky evidence first, the contract's keys LAST -- whoever keeps only the tail of a 16 KB line
        # (the driver's BENCH_rNN.json does) keeps metric / value / config / roofline / cpu_baseline / parity
        last = ("parity", "host_pixels", "ms_per_step_incl_d2h", "config", "roofline", "cpu_baseline", "metric", "value", "unit", "n_gpus", "steps", "warmup",
                "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
        ordered = {k: v for k, v in out.items() if k not in last}
        ordered.update({k: out[k] for k in last if k in out})
        print(json.dumps(ordered), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
