#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if "lt_" not in k:
                continue
            for c, v in sorted(cs.items()):
                print("%-28s %-62s n=%-3d mean=%.6g" % (c, k, len(v), sum(v) / len(v)))
