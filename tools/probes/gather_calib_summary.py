#!/usr/bin/env python3
"""FETCH_SIZE per dispatch of tools/probes/gather_calib.hip (rocprofv3 --pmc FETCH_SIZE pass) against the byte counts the
program printed, in dispatch order.   usage: gather_calib_summary.py <rocprof output dir> <program log>"""
import csv
import glob
import os
import re
import sys

rows = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            rows[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
lines = [ln for ln in open(sys.argv[2]).read().splitlines() if ln.startswith("bytes ")]
disp = [rows[k] for k in sorted(rows)]
print("%-28s %-26s %14s %16s %8s %s" % ("kernel", "table", "bytes (algo)", "FETCH_SIZE x 1 KB", "ratio", "FETCH bytes per record"))
for ln, (kname, val) in zip(lines, disp):
    m = re.search(r"bytes (\S+) table=(.+?) (?:records=(\S+) record_bytes=(\d+) )?bytes=(\S+)", ln)
    name, table, recs, rb, nbytes = m.group(1), m.group(2), m.group(3), m.group(4), float(m.group(5))
    fetched = val * 1024.0     # rocprofv3 reports FETCH_SIZE in KB
    per = "%.1f" % (fetched / float(recs)) if recs else "-"
    assert name.split("<")[0] in kname, (name, kname)
    print("%-28s %-26s %14.0f %16.0f %8.3f %s" % (name, table, nbytes, fetched, fetched / nbytes, per))
