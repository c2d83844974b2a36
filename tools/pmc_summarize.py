#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (gpurun_out/pmc/*/**/*counter_collection.csv) per kernel:
mean counter value per dispatch.  usage: python tools/pmc_summarize.py [dir] > profiles/xxx.txt"""
import re
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if "raftcorr" not in row["Kernel_Name"]:
                continue                                  # PyTorch's own kernels are not the subject
            k = re.search(r"\w+_kernel(<[^>(]*>)?", row["Kernel_Name"]).group(0)
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:34s} mean/dispatch {sum(v)/len(v):16.1f}   (n={len(v)})")
