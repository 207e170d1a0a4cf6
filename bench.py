#!/usr/bin/env python3
"""bench.py -- the headline measurement of BASELINE.json: Mrays/s (primary + secondary + shadow) and frame ms on a
synthetic 1M-triangle scene at 3840x2160, 16 samples per pixel (`accumulator` program, frameCount 1..16, device-side
running mean), image-tile-split over N GPUs of one node (one process per GPU) with ONE gather of tile pixels to
rank 0 per frame.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one complete frame of the workload: all 16 sample launches of this rank's tiles, the gather (N > 1)
and the root-side untile.  Timed region: barrier + synchronize, K steps, synchronize + barrier; MAX over ranks.
Scene and framebuffers are resident in HBM before the timed region starts.  Rank 0 prints one JSON line.

`roofline` prices the dominant kernel (one sample launch of lt_render_kernel) against HBM: achieved = ALGORITHMIC
bytes per launch / average launch duration (HIP events on the launch stream, taken inside the timed region);
algorithmic bytes = 32 B x node visits + 76 B x triangle tests + 36 B x pixels (12 B written + 24 B running-mean
read-modify-write), SURVEY.md section 8(d) -- node visits / triangle tests are the reference algorithm's counts,
measured once with device atomics (and equal to the CPU oracle's counters, see tests).
`cpu_baseline` is the CPU oracle (a C restatement of the reference kernels, kind "port") on this host's cores over
one bounded sample of the same workload; it is a reported baseline, not a target."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def measured_traffic(workload_key):
    """HBM bytes per launch of the dominant kernel from the PMC counters: they cannot be read from inside this process,
    so tools/collect_profiles.sh collects FETCH_SIZE / WRITE_SIZE for this same command in separate rocprofv3 passes
    and the summary is committed as profiles/<round>/hbm_traffic.json.  Returned only when it was measured on the same
    workload; otherwise None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "hbm_traffic.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("workload", "").startswith(workload_key):
            best = d.get("hbm_bytes_per_launch")
    return best


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--cells", type=int, default=708, help="height-field cells per side (708 -> 1 002 530 triangles)")
    ap.add_argument("--scene", default="wall", choices=["wall", "soup", "blob", "colonnade", "cornell"])
    ap.add_argument("--program", default="accumulator")
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def build_scene(args):
    from lens_trace_amd import scene as sc
    from lens_trace_amd import synth
    if args.scene == "wall":
        return synth.heightfield_wall(args.cells), "synthetic height-field wall"
    if args.scene == "soup":
        return synth.triangle_soup(2 * args.cells * args.cells), "synthetic triangle soup (seed 1)"
    if args.scene == "blob":
        return synth.blob_in_box(), "synthetic blob in a box"
    if args.scene == "colonnade":
        return synth.colonnade(), "synthetic colonnade"
    return sc.load_ltsb(os.path.join(ROOT, "tests", "golden", "cornell_box_O0.ltsb")), "Cornell box (reference buffers)"


def main():
    args = parse()
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
    mine = torch.zeros(per_rank, dtype=torch.float32, device=dev)
    # the root receives every rank's stack straight into its row of `stack` (rows are contiguous views: no staging copy)
    stack = torch.empty((world, per_rank), dtype=torch.float32, device=dev) if (world > 1 and rank == 0) else None
    gathered = [stack[i] for i in range(world)] if stack is not None else None
    image = torch.empty((H, W, D), dtype=torch.float32, device=dev) if rank == 0 else None
    stream = torch.cuda.current_stream().cuda_stream

    def step(dd):
        r.render_device(dd, mine.data_ptr(), per_rank * 4, stream)
        if world > 1:
            dist.gather(mine, gathered, dst=0)
            if rank == 0:
                r.untile(stack.data_ptr(), per_rank, world, W, H, D, plan.tile_w, plan.tile_h, image.data_ptr(), stream)

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

    # set-up, not a step: the first timed-shape call allocates the library's scratch memory (sample images of the fused
    # launch) and, under RCCL, opens the point-to-point channels of the gather
    step(d)
    for _ in range(args.warmup):
        step(d)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    shadow_walk = -1
    kernel_ms = 0.0
    render_ms = 0.0     # lt_render_kernel alone (kernel_ms also holds the running-mean kernel behind each fused launch)
    launches = 0
    for _ in range(args.steps):
        step(d)
        # the per-call HIP events sit on the launch stream; reading them waits for this step's kernels only
        s = r.stats()
        shadow_walk = s["shadow_packets"]
        kernel_ms += s["kernel_ms"]
        render_ms += s["render_ms"]
        launches += s["kernel_launches"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, kernel_ms / 1e3], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kernel_s = [float(x) for x in t.tolist()]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mrays = rays_total * args.steps / elapsed / 1e6
        launch_ms = render_ms / max(launches, 1)
        launches_per_step = max(launches, 1) / args.steps
        my_alg_bytes_per_launch = my_alg_bytes_per_step / launches_per_step
        achieved = my_alg_bytes_per_launch / (launch_ms * 1e-3) / 1e9
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
            "note": "shadow rays stop at the first accepted hit (their callers read only hitType): same pixels, fewer node visits than "
                    "the reference algorithm, whose counts (measured once with the counting kernel) price roofline.achieved",
            "config": {"workload": "%s, %d triangles, %dx%d, %d spp running mean, program %s, %s" % (
                           scene_name, scene.n_prims, W, H, args.spp, args.program,
                           "whole image on 1 GPU" if world == 1 else "%dx%d tiles interleaved over %d GPUs + 1 RCCL gather" % (plan.tile_w, plan.tile_h, world)),
                       "triangles": scene.n_prims, "bvh_nodes": scene.n_nodes, "width": W, "height": H, "spp": args.spp,
                       "rays_per_frame": rays_total, "node_visits_per_ray": nodes_total / rays_total,
                       "tri_tests_per_ray": tris_total / rays_total, "kernel_only_mrays_per_s": round(rays_total * args.steps / kernel_s / 1e6, 2),
                       "frame_ms": round(ms_per_step, 3),
                       # which of its two (pixel-identical) walks the library timed faster for this scene's shadow rays
                       "shadow_ray_walk": {1: "any-hit packets", 0: "per lane"}.get(shadow_walk, "not timed")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": measured_traffic("%s, %d triangles, %dx%d, %s" % (scene_name, scene.n_prims, W, H, args.program)) if world == 1 else None,
                         "kernel": "lt_render_kernel<%s>" % args.program if launches_per_step <= args.spp else "wavefront GI pipeline (all its stage kernels)",
                         "launch_ms": round(launch_ms, 4),
                         "launches_per_step": launches_per_step, "samples_per_launch": args.spp / launches_per_step,
                         "algorithmic_bytes_per_launch": my_alg_bytes_per_launch},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, scene, program)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


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
