#!/usr/bin/env python3
"""bench.py -- the headline measurement of BASELINE.json: Mrays/s (primary + secondary + shadow) and frame ms on a
synthetic 1M-triangle scene at 3840x2160, 16 samples per pixel (`accumulator` program, frameCount 1..16, device-side
running mean), image-tile-split over N GPUs of one node (one process per GPU) with ONE gather of tile pixels to
rank 0 per frame.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one complete frame of the workload: all 16 sample launches of this rank's tiles, the gather (N > 1)
and the root-side untile.  Timed region: barrier + synchronize, K steps, synchronize + barrier; MAX over ranks.
On several GPUs two steps are in flight (--in-flight): step k+1 is enqueued -- its own context, stream and buffers -- while step k
drains and is gathered; all K steps start and end inside the timed region.
Scene and framebuffers are resident in HBM before the timed region starts.  Rank 0 prints one JSON line.

`roofline` describes the dominant kernel (lt_render_kernel: one launch renders all samples of a step).  The kernel is not
HBM-bound -- the 1 M-triangle scene (250 MB with the derived arrays) lives in the 256 MB Infinity Cache and the eight L2s, and
the packet walks fetch a node once per wavefront through the scalar cache -- so the bound it is priced against is the busiest
instruction-issue pipe, from the rocprofv3 PMC passes of this same command (tools/collect_profiles.sh ->
profiles/<round>/issue_profile.json): `achieved` = that pipe's instructions per launch (a property of the workload and the
build) / its units / (the launch time measured LIVE here with HIP events x the profiled shader clock), `peak` = the pipe's issue
rate (MI355X_MICROARCH.md: wave64 VALU 0.5 instructions per SIMD-cycle; one scalar instruction per CU-cycle), `frac` <= 1.
Beside it: `traffic` = measured HBM bytes per launch (FETCH_SIZE + WRITE_SIZE; FETCH_SIZE doubled for streaming kernels only: profiles/r3/gather_calibration_fetch_size.txt), `hbm_measured_gbs` / `hbm_measured_frac`
against the 8 TB/s peak, and `algorithmic_gbs` = SURVEY section 8(d)'s bytes (32 B x node visits + 76 B x triangle tests +
36 B x pixel-samples, the REFERENCE algorithm's counts, measured once with device atomics and equal to the CPU oracle's) per
launch / launch time -- several times the HBM peak, which is what "on_chip_reuse_factor" states.
`cpu_baseline` is the CPU oracle (a C restatement of the reference kernels, kind "port") on this host's cores over
one bounded sample of the same workload; it is a reported baseline, not a target.

Math flavour: the library's default -- bit-identical to the reference's OpenCL kernels compiled for gfx950 the way the reference
compiles them, clBuildProgram with NULL options (src/opencl/renderer_opencl.cpp:50: a*b+c contracted inside expressions, 2.5-ulp
divide, 3-ulp sqrt).  tests/test_gpu_full_size.py compares the whole 4K frame of this workload, tests/test_gpu_reference_kernels.py
all six programs."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SHADOW_WALKS = {0: "per lane", 1: "any-hit packets", 2: "chosen per wavefront", 3: "queued for the trace kernel"}   # lt_hip_stats.shadow_packets
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


# what a launch runs besides the workload: the kernel sources and the environment switches that pick other walks / hierarchies
ENV_PINS = ("LT_RETREE", "LT_RETREE_SLACK", "LT_SHADOW_PACKETS", "LT_SHADOW_SPREAD", "LT_PERSISTENT", "LT_FUSED_FRAMES", "LT_SQUARE_MAJOR",
            "LT_NATURAL_ORDER", "LT_GI_MEGAKERNEL", "LT_GI_TRACE", "LT_GI_LDS_SCENE", "LT_TRACE_REFILL", "LT_DEBUG_LDS_ROWS")


def build_identity(ignore_pins=("LT_SHADOW_PACKETS",)):
    """sha256 over lens_trace_amd/csrc/* (what liblenstrace-hip.so is built from) and the LT_* switches in force.  Stored in
    the counter summaries under profiles/ (tools/profile_summary.py) and compared here: counters of another build say nothing
    about this one.  (LT_SHADOW_PACKETS is left out of the comparison: the profiler passes pin the walk the library picks for
    the workload by itself, so that no launch of a pass is a timing run; the summary records it.)"""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "lens_trace_amd", "csrc")
    for f in sorted(os.listdir(src)):
        h.update(f.encode())
        h.update(open(os.path.join(src, f), "rb").read())
    pins = {k: os.environ[k] for k in ENV_PINS if k in os.environ}
    return {"csrc_sha256": h.hexdigest()[:16], "env": pins, "compared_env": {k: v for k, v in pins.items() if k not in ignore_pins}}


def _latest_profile(name, workload_key):
    """profiles/r<N>/<name> of the newest round (numerically) whose `workload` is this run's AND whose build is this build
    (kernel sources and environment switches: build_identity) -- PMC counters cannot be read from inside this process:
    tools/collect_profiles.sh collects them for this same command in separate rocprofv3 passes.  Sub-directories of a round hold
    other workloads / switches and are matched the same way.  Returns (summary or None, why not)."""
    import glob
    import re
    me = build_identity()
    best, why = None, "no counter summary of this workload under profiles/"
    files = glob.glob(os.path.join(ROOT, "profiles", "r*", name)) + glob.glob(os.path.join(ROOT, "profiles", "r*", "*", name))

    def order(f):
        m = re.search(r"profiles/r(\d+)", f.replace(os.sep, "/"))
        return (int(m.group(1)) if m else -1, f.count(os.sep), f)
    for f in sorted(files, key=order):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if not d.get("workload", "").startswith(workload_key):
            continue
        b = d.get("build") or {}
        if b.get("csrc_sha256") != me["csrc_sha256"] or b.get("compared_env", {}) != me["compared_env"]:
            why = "profile stale: %s was collected from another build of the kernels (csrc %s, env %s; this build: %s, %s)" % (
                os.path.relpath(f, ROOT), b.get("csrc_sha256"), b.get("compared_env"), me["csrc_sha256"], me["compared_env"])
            continue
        best = d
        best["_file"] = os.path.relpath(f, ROOT)
    return best, (None if best else why)


def measured_traffic(workload_key):
    d, _ = _latest_profile("hbm_traffic.json", workload_key)
    return d.get("hbm_bytes_per_launch") if d else None


def roofline(workload_key, kernel_name, launch_ms, launches_per_step, spp, alg_bytes_per_launch, single_gpu):
    """See the module docstring."""
    alg_gbs = alg_bytes_per_launch / (launch_ms * 1e-3) / 1e9
    traffic = measured_traffic(workload_key) if single_gpu else None
    prof, stale = _latest_profile("issue_profile.json", workload_key) if single_gpu else (None, None)
    out = {"kernel": kernel_name, "launch_ms": round(launch_ms, 4), "launches_per_step": launches_per_step,
           "samples_per_launch": spp / launches_per_step, "traffic": traffic,
           "algorithmic_bytes_per_launch": alg_bytes_per_launch, "algorithmic_gbs": round(alg_gbs, 1)}
    if traffic:
        hbm = traffic / (launch_ms * 1e-3) / 1e9
        out.update({"hbm_measured_gbs": round(hbm, 1), "hbm_peak_gbs": HBM_PEAK_GBS, "hbm_measured_frac": round(hbm / HBM_PEAK_GBS, 4),
                    "on_chip_reuse_factor": round(alg_bytes_per_launch / traffic, 1)})
    raw = (prof or {}).get("raw", {})
    clock = (prof or {}).get("effective_clock_ghz")
    if prof and clock and "SQ_INSTS_VALU" in raw and "SQ_INSTS_SALU" in raw:
        cycles = launch_ms * 1e-3 * clock * 1e9          # live launch time x the shader clock the profiled launch held
        pipes = {"valu-issue": (raw["SQ_INSTS_VALU"] / 1024.0 / cycles, 0.5, "wave-instructions per SIMD-cycle"),
                 "scalar-issue": ((raw["SQ_INSTS_SALU"] + raw.get("SQ_INSTS_SMEM", 0.0)) / 256.0 / cycles, 1.0, "instructions per CU-cycle")}
        name = max(pipes, key=lambda k: pipes[k][0] / pipes[k][1])
        ach, peak, unit = pipes[name]
        out = dict({"bound": name, "achieved": round(ach, 4), "peak": peak, "unit": unit, "frac": round(ach / peak, 4)}, **out)
        out["pipes"] = {k: round(v[0] / v[1], 4) for k, v in pipes.items()}
        split = prof.get("wave_cycle_split") or {}
        out["wave_cycle_split"] = {k.split(" ")[0]: (round(v, 4) if v is not None else None) for k, v in split.items()}
        out["shader_clock_ghz"] = round(clock, 3)
        out["source"] = prof["_file"]
    else:
        # no counter profile of this workload AND this build in the tree: only the HBM side can be stated, if that
        hbm = (traffic / (launch_ms * 1e-3) / 1e9) if traffic else None
        out = dict({"bound": "hbm", "achieved": round(hbm, 1) if hbm else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(hbm / HBM_PEAK_GBS, 4) if hbm else None}, **out)
        if stale:
            out["note"] = stale
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--cells", type=int, default=708, help="height-field cells per side (708 -> 1 002 530 triangles)")
    ap.add_argument("--scene", default="wall", choices=["wall", "soup", "blob", "colonnade", "cornell", "mixed"])
    ap.add_argument("--program", default="accumulator")
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--bvh", default="median", choices=["median", "sah"], help="split rule of the scene's BVH (same 32-byte node layout): "
                    "the reference's median split (default, the headline) or this backend's binned SAH")
    ap.add_argument("--in-flight", type=int, default=0, choices=[0, 1, 2], help="steps in flight: 2 = step k+1 is enqueued (its own "
                    "context, stream and buffers) while step k drains and is gathered; every one of the K steps starts and ends inside "
                    "the timed region.  0 (default) = 1 on one GPU, 2 on several")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-soup", action="store_true", help="skip the second figure (config 4's incoherent triangle-soup variant)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end figures (frame into a caller's host buffer, through the plugin's render())")
    return ap.parse_args()


def build_scene(args):
    from lens_trace_amd import scene as sc
    from lens_trace_amd import synth
    sc.default_bvh = sc.BVH_SAH if args.bvh == "sah" else sc.BVH_MEDIAN
    if args.scene == "wall":
        return synth.heightfield_wall(args.cells), "synthetic height-field wall"
    if args.scene == "soup":
        return synth.triangle_soup(2 * args.cells * args.cells), "synthetic triangle soup (seed 1)"
    if args.scene == "blob":
        return synth.blob_in_box(), "synthetic blob in a box"
    if args.scene == "mixed":
        return synth.wall_and_soup(), "synthetic wall with a triangle soup in front of its left half"
    if args.scene == "colonnade":
        return synth.colonnade(), "synthetic colonnade"
    return sc.load_ltsb(os.path.join(ROOT, "tests", "golden", "cornell_box_O0.ltsb")), "Cornell box (reference buffers)"


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: this process -- before it imports torch
    or touches the GPU -- starts the N ranks as a fresh torch.distributed.run job (the command the driver uses), lets rank 0
    print the JSON line on the shared stdout, and exits with the job's status.  (Child processes, never a re-exec.)"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    import torch
    import torch.distributed as dist
    from lens_trace_amd import _capi as C
    from lens_trace_amd.dist import TilePlan
    from lens_trace_amd.renderer import RendererHIP, make_desc

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    # rehearsal knobs for a 1-GPU box: LT_BENCH_BACKEND=gloo LT_BENCH_SINGLE_DEVICE=1 runs N ranks on GPU 0 through gloo
    backend = os.environ.get("LT_BENCH_BACKEND", "nccl")
    if os.environ.get("LT_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    scene, scene_name = build_scene(args)
    if args.bvh == "sah":
        scene_name += " (binned-SAH BVH)"
    scene.validate()
    W, H, D = args.width, args.height, 3
    program = C.program_from_path(args.program)
    r = RendererHIP(local_rank)
    r.set_scene(scene)

    plan = TilePlan.balanced(W, H, D, world, args.tile)
    tile = plan.desc_tile(rank) if world > 1 else None

    def desc(stats=False):
        return make_desc(program, W, H, D, scene.camera, frame_first=1, frame_count=args.spp, accumulate=True, accumulate_base=0,
                         tile=tile, stats=stats)

    d = desc()
    my_floats = r.output_floats(d)
    # every rank's stack is padded to the largest one so that the gather is uniform
    per_rank = my_floats if world == 1 else plan.floats_per_rank
    assert my_floats <= per_rank
    # A launch ends with a drain -- the last squares' dependent chains, ~0.5 ms during which most of the chip idles -- and a rank's
    # gather and untile leave it idle too.  With two steps in flight, step k+1 (second context: its own copy of the scene, scratch,
    # stream and output buffers) fills the chip while step k drains and is gathered.  Slot = what one in-flight step owns.
    class Slot:
        pass
    if args.in_flight == 0:
        # one GPU: a 17 ms launch gains 0.6 % from overlapping its drain, and its HIP-event duration would then include the
        # neighbour's start; several GPUs: a share's launch is a few ms, a third of it drain, and the gather idles the chip
        args.in_flight = 1 if world == 1 else 2
    slots = []
    for i in range(args.in_flight):
        sl = Slot()
        sl.r = r if i == 0 else RendererHIP(local_rank)
        if i > 0:
            sl.r.set_scene(scene)
        sl.tstream = torch.cuda.current_stream() if args.in_flight == 1 else torch.cuda.Stream(device=dev)
        sl.stream = sl.tstream.cuda_stream
        sl.mine = torch.zeros(per_rank, dtype=torch.float32, device=dev)
        # the root receives every rank's stack straight into its row of `stack` (rows are contiguous views: no staging copy)
        sl.stack = torch.empty((world, per_rank), dtype=torch.float32, device=dev) if (world > 1 and rank == 0) else None
        sl.gathered = [sl.stack[j] for j in range(world)] if sl.stack is not None else None
        sl.image = torch.empty((H, W, D), dtype=torch.float32, device=dev) if rank == 0 else None
        sl.pending = False
        slots.append(sl)
    torch.cuda.synchronize()
    stream = slots[0].stream

    def step(dd, sl=None):
        sl = sl or slots[0]
        with torch.cuda.stream(sl.tstream):
            sl.r.render_device(dd, sl.mine.data_ptr(), per_rank * 4, sl.stream)
            if world > 1:
                dist.gather(sl.mine, sl.gathered, dst=0)
                if rank == 0:
                    sl.r.untile(sl.stack.data_ptr(), per_rank, world, W, H, D, plan.tile_w, plan.tile_h, sl.image.data_ptr(), sl.stream)
        sl.pending = True

    # ---- reference-algorithm work counts of this rank's share (one untimed pass with device atomics) ----
    step(desc(stats=True))
    torch.cuda.synchronize()
    st = r.stats()
    counts = torch.tensor([st["rays"], st["shadow_rays"], st["node_visits"], st["tri_tests"], st["pixels"]], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(counts)
    rays_total, shadow_total, nodes_total, tris_total, pixels_total = [float(x) for x in counts.tolist()]
    # algorithmic bytes of one step of this rank (32 B per node visit, 76 B per triangle test, 12 B per pixel
    # written + 24 B read-modify-write per accumulated sample), divided below by the render launches the step really made
    # (one launch per sample, or ONE for all samples when the library fuses them: lt_capi.hip, render_on_stream)
    my_alg_bytes_per_step = 32.0 * st["node_visits"] + 76.0 * st["tri_tests"] + 36.0 * st["pixels"] * args.spp

    # set-up, not a step: the first timed-shape call of a context allocates the library's scratch memory (sample images of the
    # fused launch), times the shadow-ray walks once and, under RCCL, opens the point-to-point channels of the gather
    for sl in slots:
        step(d, sl)
        torch.cuda.synchronize()
    # Several steps in flight: the library timed the shadow-ray walks on ONE launch at a time (first call, above), and a launch alone
    # pays its drain in full -- on a 1/8 share any-hit packets lose that timing to the per-lane walk (3.1 against 2.6 ms) and win
    # once the next step's wavefronts fill the drain (2.24 against 2.50 ms per step).  Whoever pipelines knows: the candidates
    # (the library's verdict, packets, per lane) are timed the way the steps will run, every rank takes the walk that is fastest
    # for the slowest rank, and it stays pinned (LT_SHADOW_PACKETS) for the warm-up and the timed region.  Set-up, not a step.
    pipelined_choice = None
    if len(slots) > 1 and args.program == "accumulator" and "LT_SHADOW_PACKETS" not in os.environ:
        candidates = [None, "1", "0"]
        took = []
        for cand in candidates:
            if cand is None:
                os.environ.pop("LT_SHADOW_PACKETS", None)
            else:
                os.environ["LT_SHADOW_PACKETS"] = cand
            for k in range(2):
                step(d, slots[k % len(slots)])
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            tq = time.perf_counter()
            for k in range(8):
                step(d, slots[k % len(slots)])
            torch.cuda.synchronize()
            took.append((time.perf_counter() - tq) / 8 * 1e3)
        tt = torch.tensor(took, dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        took = [float(x) for x in tt.tolist()]
        best = min(range(len(candidates)), key=lambda i: took[i])
        if candidates[best] is None:
            os.environ.pop("LT_SHADOW_PACKETS", None)
        else:
            os.environ["LT_SHADOW_PACKETS"] = candidates[best]
        pipelined_choice = {"ms_per_step_library_verdict": round(took[0], 3), "ms_per_step_packets": round(took[1], 3),
                            "ms_per_step_per_lane": round(took[2], 3), "pinned": candidates[best] or "library verdict"}
    for k in range(args.warmup):
        step(d, slots[k % len(slots)])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    own_tree = (r.stats()["own_tree_height"], r.stats()["own_tree_ms"])
    for sl in slots:
        sl.pending = False
    t0 = time.perf_counter()
    shadow_walk = -1
    kernel_ms = 0.0
    render_ms = 0.0     # lt_render_kernel alone (kernel_ms also holds the running-mean kernel behind each fused launch)
    launches = 0

    def collect(sl):
        # the per-call HIP events sit on the launch stream; reading them waits for that step's kernels only
        nonlocal shadow_walk, kernel_ms, render_ms, launches
        st_ = sl.r.stats()
        shadow_walk = st_["shadow_packets"]
        kernel_ms += st_["kernel_ms"]
        render_ms += st_["render_ms"]
        launches += st_["kernel_launches"]
        sl.pending = False

    for k in range(args.steps):
        sl = slots[k % len(slots)]
        if sl.pending:
            collect(sl)     # (the step that used this slot before: at most len(slots) steps are ever in flight)
        step(d, sl)
    for sl in slots:
        if sl.pending:
            collect(sl)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, kernel_ms / 1e3], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kernel_s = [float(x) for x in t.tolist()]
    if pipelined_choice:
        os.environ.pop("LT_SHADOW_PACKETS", None)     # (what follows runs one step at a time again: the library's own verdict)

    # The latency of ONE frame, outside the timed region: a few steps one at a time (enqueue, gather, untile, synchronize, barrier),
    # MAX over ranks.  With one step in flight this is what ms_per_step is; with two, ms_per_step is the pipelined period and a frame
    # takes longer than that from its first launch to its last pixel.
    lat_steps = max(2, min(args.steps, 5))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    lat_render_ms, lat_launches = 0.0, 0
    for _ in range(lat_steps):
        step(d, slots[0])
        torch.cuda.synchronize()
        st_ = slots[0].r.stats()
        lat_render_ms += st_["render_ms"]
        lat_launches += st_["kernel_launches"]
        if world > 1:
            dist.barrier()
    lat = torch.tensor([(time.perf_counter() - t0) / lat_steps], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(lat, op=dist.ReduceOp.MAX)
    frame_latency_ms = float(lat.item()) * 1e3
    if args.in_flight > 1:     # (HIP-event times of overlapping steps include the neighbour's work: the kernel's own time comes from these steps)
        render_ms, launches = lat_render_ms * args.steps / lat_steps, int(round(lat_launches * args.steps / lat_steps))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mrays = rays_total * args.steps / elapsed / 1e6
        launch_ms = render_ms / max(launches, 1)
        launches_per_step = max(launches, 1) / args.steps
        my_alg_bytes_per_launch = my_alg_bytes_per_step / launches_per_step
        out = {
            # BASELINE.json's metric on its configuration; other --scene / --width / --height runs say what they ran
            "metric": "Mrays/s (primary+secondary+shadow), %s @%s" % (
                "1M-tri" if 990000 <= scene.n_prims <= 1010000 else "%d-tri" % scene.n_prims, "4K" if (W, H) == (3840, 2160) else "%dx%d" % (W, H)),
            "value": round(mrays, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "note": "math flavour = the library default, bit-identical to the reference's OpenCL kernels built for gfx950 as the "
                    "reference builds them (NULL build options, renderer_opencl.cpp:50) -- whole 4K frame of this workload in "
                    "tests/test_gpu_full_size.py; "
                    "finite rays walk the backend's own binned-SAH hierarchy over the caller's leaves (built at set_scene; a ray "
                    "reaches the same leaves through any nested hierarchy: lens_trace_amd/csrc/lt_retree.hpp) and shadow rays stop at "
                    "the first accepted hit (their callers read only hitType): same pixels, fewer node visits than the reference "
                    "algorithm, whose ray / node / triangle counts (measured once with the counting kernel, which walks the caller's "
                    "tree in the reference's order) define the rays of `value` and price roofline.algorithmic_gbs; "
                    "config.callers_splits_* = the same walks over the caller's own splits (LT_RETREE=0)",
            "config": {"workload": "%s, %d triangles, %dx%d, %d spp running mean, program %s, %s" % (
                           scene_name, scene.n_prims, W, H, args.spp, args.program,
                           "whole image on 1 GPU" if world == 1 else "%dx%d tiles interleaved over %d GPUs + 1 RCCL gather" % (plan.tile_w, plan.tile_h, world)),
                       "workload_key": "%s, %d triangles, %dx%d, %s" % (scene_name, scene.n_prims, W, H, args.program),   # (names the profiles/ summaries of this workload)
                       "triangles": scene.n_prims, "bvh_nodes": scene.n_nodes, "bvh_split": args.bvh, "width": W, "height": H, "spp": args.spp,
                       "rays_per_frame": rays_total, "node_visits_per_ray": nodes_total / rays_total,
                       "tri_tests_per_ray": tris_total / rays_total,
                       # (per-call HIP-event times include the neighbouring step's work when two steps overlap: not reported then)
                       "kernel_only_mrays_per_s": round(rays_total * args.steps / kernel_s / 1e6, 2) if args.in_flight == 1 else None,
                       # frame ms, both ways: the period at which finished frames leave the job (= ms_per_step; with two steps in
                       # flight, two frames overlap) and the latency of one frame rendered alone, first launch to last pixel at the root
                       "frame_period_ms": round(ms_per_step, 3), "frame_latency_ms": round(frame_latency_ms, 3), "steps_in_flight": args.in_flight,
                       # which of its three (pixel-identical) walks the library timed fastest for this scene's shadow rays
                       "shadow_ray_walk": SHADOW_WALKS.get(shadow_walk, "not timed"),
                       # (steps in flight > 1: the walks timed the way the steps run, see above)
                       **({"shadow_ray_walk_pipelined": pipelined_choice} if pipelined_choice else {}),
                       # the backend's own hierarchy over the caller's leaves: height, time of its preparation at set_scene (not in any step)
                       "own_hierarchy_height": own_tree[0], "own_hierarchy_build_ms": round(own_tree[1], 1)},
            "roofline": roofline("%s, %d triangles, %dx%d, %s" % (scene_name, scene.n_prims, W, H, args.program),
                                 "lt_render_kernel<%s>" % args.program if launches_per_step <= args.spp else "wavefront GI pipeline (all its stage kernels)",
                                 launch_ms, launches_per_step, args.spp, my_alg_bytes_per_launch, world == 1),
        }
        if world == 1 and not args.no_e2e:
            out["config"].update(e2e_figures(args, scene, r))
        if world == 1 and args.scene == "wall" and not args.no_soup:
            # the same walks over the caller's own (median-split) hierarchy ...
            out["config"].update(callers_splits_figure(args, scene, program, dev, stream, rays_total))
            # ... and: the headline scene is the coherent one; the same kernel on config 4's seeded triangle-soup variant beside it
            out["config"].update(soup_figure(args, r, program, dev, stream))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, scene, program)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for sl in slots[1:]:
        sl.r.close()
    r.close()


def e2e_figures(args, scene, r):
    """The same frame END TO END, the way the reference's contract has it -- on return the caller's HOST buffer is complete
    (src/opencl/renderer_opencl.cpp:146-149) -- a few whole steps each, after the timed region:
      e2e_frame_ms_host_buffer : lt_hip_render: the step + the 99.5 MB read-back (pinned pieces, copied on by host threads)
      e2e_frame_ms_plugin      : RendererHIP.render() with the default scene-change contract, i.e. the caller hands over its scene
                                 with every frame as the reference's callers do (renderer_opencl.cpp:107-120): lt_hip_render_scene
                                 hashes all of it while the frame renders
      scene_hash_ms            : that hash alone (lt_hip_set_scene of the unchanged scene), for scale: it is not added to anything
      scene_change_ms          : lt_hip_set_scene of the same scene with ANOTHER node buffer (the root's box a little larger): hash,
                                 upload, checks, leaf order table, own hierarchy, 4-wide groups, walk records -- everything anew;
                                 scene_prepared_on says whether kernels (lt_prep.hip) or host threads (lt_retree.hpp) made them
      e2e_frame_ms_plugin_changing_scene : RendererHIP.render() of a caller whose node buffer differs from the last call's every time"""
    import numpy as np
    from lens_trace_amd.renderer import RenderPropertiesHIP
    W, H = args.width, args.height
    out = np.empty((H, W, 3), dtype=np.float32)
    path = args.program if "/" in args.program or "." in args.program else args.program + ".cl"
    if args.program == "global_illumination":
        path = "examples/global_illumination/resources/kernels/global_illumination.cl"
    steps = max(2, min(args.steps, 5))

    def timed(props):
        r.render(props)
        t0 = time.perf_counter()
        for _ in range(steps):
            r.render(props)
        return (time.perf_counter() - t0) / steps * 1e3
    base = dict(kernelFilePath=path, imageDimensions=(W, H, 3), pOutputBuffer=out, pAccelerationStructureExplicit=scene, pCamera=scene.camera,
                frameFirst=1, frameCount=args.spp, accumulate=True)
    r.set_scene(scene)
    versioned = timed(RenderPropertiesHIP(sceneVersion=1, **base))      # the scene is known: lt_hip_render
    plugin = timed(RenderPropertiesHIP(sceneVersion=0, **base))         # the scene comes with the frame: lt_hip_render_scene
    t0 = time.perf_counter()
    for _ in range(steps):
        r.set_scene(scene)
    hash_ms = (time.perf_counter() - t0) / steps * 1e3
    change, keep = [], []     # (keep: the scenes stay alive, so that no 64 MB munmap of the one before lands in the timed call)
    for k in range(steps + 1):
        nodes = scene.node_view.copy()
        nodes["boundsMax"][0][0] += 1e-3 * (k + 1)
        other = type(scene)(nodes=nodes.view(np.uint8).reshape(-1), prims=scene.prims, materials=scene.materials, lights=scene.lights, camera=scene.camera)
        keep.append(other)
        t0 = time.perf_counter()
        r.set_scene(other)
        change.append((time.perf_counter() - t0) * 1e3)
    info = r.scene_structure(3)
    # ... and whole frames of a caller whose node buffer is another one every call (an animation that rebuilds its tree): hash,
    # preparation, frame, read-back
    moving = []
    for k in range(steps + 1):
        props = RenderPropertiesHIP(sceneVersion=0, **dict(base, pAccelerationStructureExplicit=keep[k % len(keep)]))
        t0 = time.perf_counter()
        r.render(props)
        moving.append((time.perf_counter() - t0) * 1e3)
    r.set_scene(scene)
    return {"e2e_frame_ms_host_buffer": round(versioned, 3), "e2e_frame_ms_plugin": round(plugin, 3), "scene_hash_ms": round(hash_ms, 3),
            "scene_change_ms": round(min(change[1:]), 3), "scene_prepared_on": "device" if info[3] else "host",
            "e2e_frame_ms_plugin_changing_scene": round(sum(moving[1:]) / len(moving[1:]), 3),
            "e2e_readback_bytes": int(out.nbytes), "e2e_scene_bytes": int(scene.nodes.nbytes + scene.prims.nbytes + scene.materials.nbytes + scene.lights.nbytes)}


def callers_splits_figure(args, scene, program, dev, stream, rays_per_step):
    """The same workload with LT_RETREE=0: the walks' structures are built over the caller's own splits (the reference builder's
    median splits for the default --bvh) instead of the backend's binned-SAH splits.  Same pixels; a few whole steps."""
    import torch
    from lens_trace_amd.renderer import RendererHIP, make_desc
    old = os.environ.get("LT_RETREE")
    os.environ["LT_RETREE"] = "0"
    try:
        r2 = RendererHIP(0)
        r2.set_scene(scene)
    finally:
        if old is None:
            del os.environ["LT_RETREE"]
        else:
            os.environ["LT_RETREE"] = old
    W, H, D = args.width, args.height, 3
    d = make_desc(program, W, H, D, scene.camera, frame_first=1, frame_count=args.spp, accumulate=True, accumulate_base=0)
    buf = torch.zeros(r2.output_floats(d), dtype=torch.float32, device=dev)
    for _ in range(2):     # allocates scratch, times the shadow-ray walks once, warms up
        r2.render_device(d, buf.data_ptr(), buf.numel() * 4, stream)
    torch.cuda.synchronize()
    steps = max(2, min(args.steps, 5))
    t0 = time.perf_counter()
    for _ in range(steps):
        r2.render_device(d, buf.data_ptr(), buf.numel() * 4, stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = r2.stats()
    r2.close()
    return {"callers_splits_mrays_per_s": round(rays_per_step * steps / dt / 1e6, 2), "callers_splits_frame_ms": round(dt / steps * 1e3, 3),
            "callers_splits_height": st["own_tree_height"]}


def soup_figure(args, r, program, dev, stream):
    """Config 4's incoherent variant (seeded triangle soup, same triangle count, resolution, program and samples):
    Mrays/s of a few whole steps on this GPU, rays counted by the counting kernel."""
    import torch
    from lens_trace_amd import synth
    from lens_trace_amd.renderer import make_desc
    soup = synth.triangle_soup(2 * args.cells * args.cells).validate()
    r.set_scene(soup)
    W, H, D = args.width, args.height, 3

    def desc(stats=False):
        return make_desc(program, W, H, D, soup.camera, frame_first=1, frame_count=args.spp, accumulate=True, accumulate_base=0, stats=stats)
    buf = torch.zeros(r.output_floats(desc()), dtype=torch.float32, device=dev)
    r.render_device(desc(True), buf.data_ptr(), buf.numel() * 4, stream)
    torch.cuda.synchronize()
    st = r.stats()
    for _ in range(2):     # allocates scratch, times the two shadow-ray walks once, warms up
        r.render_device(desc(), buf.data_ptr(), buf.numel() * 4, stream)
    torch.cuda.synchronize()
    steps = max(2, min(args.steps, 5))
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render_device(desc(), buf.data_ptr(), buf.numel() * 4, stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    walk = r.stats()["shadow_packets"]
    return {"soup_mrays_per_s": round(st["rays"] * steps / dt / 1e6, 2), "soup_frame_ms": round(dt / steps * 1e3, 3),
            "soup_triangles": soup.n_prims, "soup_node_visits_per_ray": st["node_visits"] / st["rays"],
            "soup_shadow_ray_walk": SHADOW_WALKS.get(walk, "not timed")}


def cpu_baseline(args, scene, program):
    """The CPU oracle on this host's cores, over a bounded sample of the same workload: rows of frame 1."""
    from lens_trace_amd import scene as sc
    from oracle import pyoracle as po
    cores = len(os.sched_getaffinity(0))
    W, H = args.width, args.height
    # calibrate on frame 1, then render whole frames (frameCount 1, 2, ...) for ~15 s of CPU work
    t0 = time.perf_counter()
    _, st = po.render(scene, sc.camera_with_frame(scene.camera, 1), W, H, program, threads=cores, want_stats=True)
    dt1 = max(time.perf_counter() - t0, 1e-3)
    frames = int(max(1, min(args.spp, round(15.0 / dt1))))
    rays = 0
    t0 = time.perf_counter()
    for f in range(1, frames + 1):
        _, st = po.render(scene, sc.camera_with_frame(scene.camera, f), W, H, program, threads=cores, want_stats=True)
        rays += st["rays"]
    dt = time.perf_counter() - t0
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d whole frames (frameCount 1..%d) of the workload, %dx%d, %d rays in %.1f s on %d threads" % (
                frames, frames, W, H, rays, dt, cores)}


if __name__ == "__main__":
    main()
