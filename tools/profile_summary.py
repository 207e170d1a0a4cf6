#!/usr/bin/env python3
"""Turns the rocprofv3 passes of tools/collect_profiles.sh into the two summaries bench.py reads:

  hbm_traffic.json   : HBM bytes per launch of the dominant kernel = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes; the factor 2 is
                       MI355X_MICROARCH.md's gfx950 correction for wide coalesced reads)
  issue_profile.json : what the kernel's wave-cycles are spent on and how busy each issue pipe is, against the peaks of
                       MI355X_MICROARCH.md (wave64 VALU: 2 cycles per instruction per SIMD-32, so 0.5 wave-instructions per
                       SIMD-cycle; one scalar instruction per CU-cycle)

usage: profile_summary.py <gpurun_out/prof_rN> "<command the passes profiled>" """
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out_dir = sys.argv[1]
command = sys.argv[2] if len(sys.argv) > 2 else ""
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_identity   # noqa: E402  (hash of the kernel sources + the environment pins that change what a launch runs)


def counters(name):
    """{kernel: {counter: mean per dispatch}} of one pass, and {kernel: dispatches}"""
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out_dir, name, "*", "*_counter_collection.csv")):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        dispatches[k] = max(dispatches.get(k, 0), max(len(v) for v in cs.values()))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


dispatches = {}


def dominant(per_kernel, counter):
    """the non-counting render kernel (or GI stage set) with the largest total of `counter`"""
    best = None
    for k, cs in per_kernel.items():
        m = re.search(r"Config<(\w+), (\w+), (\w+)(?:, (\w+))?>", k)
        if "lt_" not in k or (m and m.group(2) == "true"):   # (Config<DEEP, STATS, DEVLIBM, LDSSCENE>: skip the counting instantiations)
            continue
        if counter in cs and (best is None or cs[counter] * dispatches.get(k, 1) > per_kernel[best][counter] * dispatches.get(best, 1)):
            best = k      # (the largest TOTAL over the run: mean per dispatch x dispatches)
    return best


passes = {n: counters(n) for n in ("fetch", "write", "issue", "pipes", "insts", "sqc", "cache")}
bench = {}
try:
    bench = json.loads([ln for ln in open(os.path.join(out_dir, "stats.log")).read().splitlines() if ln.startswith('{"metric"')][-1])
except Exception:
    pass

# kernel duration from the --stats pass
dur_ms = {}
for f in glob.glob(os.path.join(out_dir, "stats", "*", "*_kernel_stats.csv")):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            dur_ms[row["Name"]] = (float(row["AverageNs"]) / 1e6, int(row["Calls"]))

kern = dominant(passes["issue"], "SQ_WAVE_CYCLES") or dominant(passes["fetch"], "FETCH_SIZE")
if kern is None:
    sys.exit("no lt_ kernel found in the counter passes")
ms, calls = dur_ms.get(kern, (None, 0))
# bench.py's workload key: "<scene name>, <n> triangles, <W>x<H>, <program>"
key = (bench.get("config") or {}).get("workload_key", "")

if kern in passes["fetch"] and kern in passes["write"]:
    fetch_kb, write_kb = passes["fetch"][kern]["FETCH_SIZE"], passes["write"][kern]["WRITE_SIZE"]
    # FETCH_SIZE counts fabric read requests at 64 bytes each.  Calibrated on this chip (tools/probes/gather_calib.hip,
    # profiles/r3/gather_calibration.txt): a wide coalesced stream moves 128 bytes per request (the guide's x2), but the reads of
    # these kernels -- per-lane gathers of 16..64-byte records and wave-uniform s_load_dwordx16 -- are counted at 62-64 bytes per
    # 64-byte record: x1.  Only the running-mean kernel streams; it is not the dominant kernel.
    streaming = "running_mean" in kern or "untile" in kern
    factor = 2 if streaming else 1
    hbm = {"command": command, "kernel": kern, "workload": key, "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
           "correction": "FETCH_SIZE = fabric read requests x 64 B; x%d for this kernel (gathers of <= 64-byte records and scalar loads are counted as "
                         "they are, wide coalesced streams at half: tools/probes/gather_calib.hip); WRITE_SIZE as reported" % factor,
           "hbm_bytes_per_launch": int(factor * fetch_kb * 1024 + write_kb * 1024), "launch_ms_under_profiler": ms, "build": build_identity()}
    json.dump(hbm, open(os.path.join(out_dir, "hbm_traffic.json"), "w"), indent=1)
    print("hbm_traffic.json:", hbm["hbm_bytes_per_launch"] / 1e9, "GB per launch")

def profile_of(kern):
    c = {}
    for n in ("issue", "pipes", "insts", "sqc", "cache"):
        c.update(passes[n].get(kern, {}))
    if "SQ_WAVE_CYCLES" not in c:
        return None
    ms = dur_ms.get(kern, (None, 0))[0]
    if True:
        CUS, SIMDS = 256, 1024
        # Shader cycles of one launch: SQ_BUSY_CYCLES is summed over the chip's 32 shader engines (MI355X_MICROARCH.md, chip-level
        # parameters), so SQ_BUSY_CYCLES / 32 is the time the launch kept the shader busy, in shader clocks; divided by the
        # launch's duration it gives the clock the chip held (2.36-2.39 GHz here).  (GRBM_GUI_ACTIVE / 8, the guide's DVFS
        # recipe, reads ~25 % high on this persistent kernel under per-dispatch counter collection, so it is not used.)
        # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md, cycle constants).
        cycles = c.get("SQ_BUSY_CYCLES", 0) / 32.0
        if not cycles and ms:
            cycles = ms * 1e-3 * 2.4e9
        wave_quads = c["SQ_WAVE_CYCLES"]
        prof = {"command": command, "kernel": kern, "workload": key, "build": build_identity(), "launch_ms_under_profiler": ms, "shader_cycles_per_launch": cycles,
                "effective_clock_ghz": (cycles / (ms * 1e-3) / 1e9) if ms else None, "raw": c}
        frac = lambda x: (c[x] / wave_quads) if x in c else None
        prof["wave_cycle_split"] = {"waiting_on_memory_or_barrier (SQ_WAIT_ANY)": frac("SQ_WAIT_ANY"),
                                    "issue_stalled (SQ_WAIT_INST_ANY)": frac("SQ_WAIT_INST_ANY"),
                                    "executing (SQ_ACTIVE_INST_ANY)": frac("SQ_ACTIVE_INST_ANY")}
        if cycles:
            pipes = {}
            if "SQ_INSTS_VALU" in c:
                pipes["valu"] = {"wave_instructions_per_simd_cycle": c["SQ_INSTS_VALU"] / SIMDS / cycles, "peak": 0.5,
                                 "frac": c["SQ_INSTS_VALU"] / SIMDS / cycles / 0.5}
            if "SQ_INSTS_SALU" in c:
                n = c["SQ_INSTS_SALU"] + c.get("SQ_INSTS_SMEM", 0.0)
                pipes["scalar"] = {"instructions_per_cu_cycle (SALU + SMEM)": n / CUS / cycles, "peak": 1.0, "frac": n / CUS / cycles,
                                   "salu_only_frac": c["SQ_INSTS_SALU"] / CUS / cycles}
            if "SQ_INSTS_VMEM_RD" in c:
                pipes["vmem"] = {"instructions_per_cu_cycle": (c["SQ_INSTS_VMEM_RD"] + c.get("SQ_INSTS_VMEM_WR", 0)) / CUS / cycles}
            if "SQ_INSTS_LDS" in c:
                pipes["lds"] = {"instructions_per_cu_cycle": c["SQ_INSTS_LDS"] / CUS / cycles}
            if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
                pipes["valu"]["lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
            if "TA_BUSY_avr" in c:
                pipes["texture_addresser_busy"] = c["TA_BUSY_avr"] / cycles
            prof["pipes"] = pipes
        if "SQC_ICACHE_REQ" in c:
            prof["instruction_cache"] = {"requests": c["SQC_ICACHE_REQ"], "hit_rate": c["SQC_ICACHE_HITS"] / max(c["SQC_ICACHE_REQ"], 1.0),
                                         "misses": c["SQC_ICACHE_MISSES"], "misses_per_1000_instructions":
                                             1000.0 * c["SQC_ICACHE_MISSES"] / max(c.get("SQ_INSTS_VALU", 0) + c.get("SQ_INSTS_SALU", 0), 1.0)}
        if "SQC_DCACHE_REQ" in c:
            prof["scalar_cache"] = {"requests": c["SQC_DCACHE_REQ"], "hit_rate": c["SQC_DCACHE_HITS"] / max(c["SQC_DCACHE_REQ"], 1.0),
                                    "misses": c["SQC_DCACHE_MISSES"]}
        if "TCC_HIT_sum" in c:
            prof["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0)
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
            prof["l1_hit_rate"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / max(c["TCP_TOTAL_CACHE_ACCESSES_sum"], 1.0)
        # the binding class: the busiest issue pipe, unless the waves spend most of their cycles parked on memory
        cand = {k: v["frac"] for k, v in prof.get("pipes", {}).items() if isinstance(v, dict) and "frac" in v}
        if cand:
            top = max(cand, key=cand.get)
            prof["bound"] = {"class": {"valu": "valu-issue", "scalar": "scalar-issue"}[top], "frac": cand[top], "all": cand}
        prof["dispatches_profiled"] = dispatches.get(kern)
        return prof


prof = profile_of(kern)
if prof:
    # every other lt_ kernel of the run beside the dominant one (the stage kernels of the GI pipeline, the running mean, ...)
    others = {}
    for k in sorted(passes["issue"]):
        if "lt_" in k and k != kern:
            po = profile_of(k)
            if po:
                po.pop("raw", None)
                others[k] = po
    prof["other_kernels"] = others
    json.dump(prof, open(os.path.join(out_dir, "issue_profile.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in prof.items() if k not in ("raw", "other_kernels")}, indent=1))
