"""GPU test (-m gpu) of bench.py's output contract on a reduced workload: one JSON line with the driver's keys, the roofline
and cpu_baseline objects, and the two-rank path (two processes sharing this GPU through gloo: LT_BENCH_BACKEND=gloo,
LT_BENCH_SINGLE_DEVICE=1) producing the same ray count as one rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--cells", "64", "--width", "640", "--height", "360", "--spp", "4", "--steps", "2", "--warmup", "1"]


def run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run(cmd, cwd=ROOT, env=e, check=True, capture_output=True, text=True, timeout=600).stdout
    lines = [line for line in out.splitlines() if line.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_keys():
    d = run([sys.executable, "bench.py", "--gpus", "1"] + SMALL)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "Mrays/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["config"]["spp"] == 4
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_gbs", "launch_ms"):
        assert key in r, key
    assert r["launches_per_step"] == 1.0 and r["samples_per_launch"] == 4.0      # all samples of a step in one launch
    # no counter profile exists for this reduced workload: nothing measured, nothing claimed
    assert r["traffic"] is None and r["frac"] is None and r["bound"] == "hbm" and r["peak"] == 8000.0
    assert r["algorithmic_gbs"] > 0
    assert "soup_mrays_per_s" in d["config"] and d["config"]["soup_mrays_per_s"] > 0
    # end to end (the reference's contract: the caller's host buffer is complete on return), and one frame's latency
    cfg = d["config"]
    assert cfg["e2e_frame_ms_host_buffer"] > 0 and cfg["e2e_frame_ms_plugin"] > 0 and cfg["scene_hash_ms"] > 0
    assert cfg["frame_latency_ms"] > 0 and cfg["frame_period_ms"] == d["ms_per_step"] and cfg["steps_in_flight"] == 1
    assert cfg["scene_change_ms"] > cfg["scene_hash_ms"] and cfg["scene_prepared_on"] in ("device", "host")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "frames" in c["sample"]
    assert d["value"] > c["value"]
    # rays = one camera ray per pixel + one shadow ray per hit that is not a light, per sample
    px = 640 * 360 * 4
    assert px <= d["config"]["rays_per_frame"] <= 2 * px


def test_two_ranks_on_one_gpu_through_gloo_count_the_same_rays():
    one = run([sys.executable, "bench.py", "--gpus", "1", "--no-cpu-baseline"] + SMALL)
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29541", "bench.py", "--gpus", "2", "--no-cpu-baseline"] + SMALL,
              env={"LT_BENCH_BACKEND": "gloo", "LT_BENCH_SINGLE_DEVICE": "1"})
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    # two steps in flight: the period of finished frames and one frame's own latency are two numbers
    assert two["config"]["steps_in_flight"] == 2 and two["config"]["frame_latency_ms"] > 0 and two["config"]["kernel_only_mrays_per_s"] is None
    # ... and the shadow-ray walk was chosen the way the steps run (two in flight), not from one launch alone
    pc = two["config"]["shadow_ray_walk_pipelined"]
    assert pc["pinned"] in ("library verdict", "1", "0") and min(pc["ms_per_step_packets"], pc["ms_per_step_per_lane"], pc["ms_per_step_library_verdict"]) > 0
    assert two["config"]["rays_per_frame"] == one["config"]["rays_per_frame"]
    assert "tiles interleaved over 2 GPUs" in two["config"]["workload"]
    assert "cpu_baseline" not in two


def test_plain_invocation_with_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` as the driver words the N = 1 command: no torch.distributed environment -- bench.py starts the
    ranks itself (child processes) and relays rank 0's line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"LT_BENCH_BACKEND": "gloo", "LT_BENCH_SINGLE_DEVICE": "1"})
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--no-cpu-baseline"] + SMALL, cwd=ROOT, env=env, check=True,
                         capture_output=True, text=True, timeout=600).stdout
    lines = [line for line in out.splitlines() if line.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "tiles interleaved over 2 GPUs" in d["config"]["workload"]


def test_full_workload_line_prices_the_built_kernel_or_says_why_not():
    """The headline workload itself (a few steps, no side figures): `roofline.frac` is a fraction of a stated issue peak taken from the
    counter summary under profiles/ whose `build` is THIS build of the kernels (bench.build_identity), with the measured HBM side
    beside it -- or it is null and `roofline.note` says that the committed counters belong to another build.  Never a number
    computed from stale counters."""
    d = run([sys.executable, "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-soup", "--no-e2e"])
    r = d["roofline"]
    assert d["config"]["triangles"] == 1002530 and d["config"]["width"] == 3840 and d["value"] > 5000
    if r["frac"] is None:
        assert "stale" in r.get("note", ""), r
    else:
        assert r["bound"] in ("valu-issue", "scalar-issue") and 0.0 < r["frac"] <= 1.0
        assert 0.0 < r["hbm_measured_frac"] <= 1.0 and r["traffic"] > 0 and r["source"].startswith("profiles/r")
        # SURVEY 8(d)'s bytes of the reference algorithm per launch exceed what the chip can move: the kernel does not move them
        assert r["algorithmic_gbs"] > r["hbm_peak_gbs"] and r["on_chip_reuse_factor"] > 1.0
