#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the rocprofv3 PMC passes (tools/pmc_passes.sh): per-launch HBM
bytes per kernel = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024.  The factor 2 is the gfx950
correction (MI355X_MICROARCH.md, HBM section; re-measured for sparse reads in
profiles/r01_fetch_calibration.txt): FETCH_SIZE tallies each 128-B request at 64 B."""
import re, csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
dest = sys.argv[2] if len(sys.argv) > 2 else "profiles/pmc_traffic.json"
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if "raftcorr" not in row["Kernel_Name"]:
            continue
        k = re.search(r"\w+_kernel", row["Kernel_Name"]).group(0)
        if "lookup_fwd_kernel" in k and re.search(r"lookup_fwd_kernel<-?\d+, [12]>", row["Kernel_Name"]):
            k += "_16bit_volume"
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, c in acc.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]); w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
        out[k] = round(2 * f * 1024 + w * 1024)
json.dump(out, open(dest, "w"), indent=1)
print(out)
