#!/usr/bin/env python3
"""Machine-code size, registers, scratch and occupancy of every kernel of liblenstrace-hip.so, from the gfx950 assembly:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -Iinclude -Ilens_trace_amd/csrc \
        lens_trace_amd/csrc/lt_capi.hip -o /tmp/lt_capi.s && python3 tools/isa_size.py /tmp/lt_capi.s > profiles/<round>/isa_size.txt"""
import re
import subprocess
import sys

rows = []
name = None
cur = {}
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+|lt_\w+):\s", line)
    if m:
        name, cur = m.group(1), {}
        continue
    m = re.match(r"^; (codeLenInByte|NumVgprs|TotalNumSgprs|ScratchSize|Occupancy)\s*[:=]\s*(\d+)", line)
    if m and name:
        cur[m.group(1)] = int(m.group(2))
        if m.group(1) == "Occupancy" and "codeLenInByte" in cur:
            rows.append((cur["codeLenInByte"], cur.get("NumVgprs", 0), cur.get("TotalNumSgprs", 0), cur.get("ScratchSize", 0), cur["Occupancy"], name))
            name = None
names = subprocess.run(["c++filt"], input="\n".join(r[5] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("# machine-code size of every kernel of liblenstrace-hip.so (tools/isa_size.py on the gfx950 assembly of lt_capi.hip and lt_prep.hip), largest first:")
print("# bytes, VGPRs, SGPRs, scratch bytes per lane, waves per SIMD.  The instruction cache two CUs share holds 64 KB; the")
print("# instruction-cache hit rate of the bench launch is in issue_profile.json (0.999997): the walks' loops are small and hot.")
for (size, vg, sg, scratch, occ, _), n in sorted(zip(rows, names), key=lambda t: -t[0][0]):
    n = re.sub(r"\(.*$", "", n)
    print("%7d bytes  vgpr %3d  sgpr %3d  scratch %4d B/lane  waves/SIMD %d  %s" % (size, vg, sg, scratch, occ, n))
