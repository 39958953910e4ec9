#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter_collection.csv files to one row per (sufr kernel, counter):
the value of the dispatch with the LARGEST grid (the top-level launch over the whole genome), the mean over all
its dispatches and the largest value of any dispatch (kernels whose grid is fixed: the heavy launch)."""
import csv
import glob
import sys
from collections import defaultdict

import hashlib
import os

if len(sys.argv) < 3 or any(a.startswith("-") for a in sys.argv[1:]):
    # (an output path that starts with "-" once left a file named "--help" in the repository root)
    sys.exit("usage: summarize_pmc.py <out.csv> <rocprofv3 output dir> [...]   (paths must not start with '-')")
out, dirs = sys.argv[1], sys.argv[2:]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash(name="sufr_part.inc"):
    """sha256 of the kernel source the counters belong to: bench.py refuses the file's traffic figure when the source it
    runs differs (evidence that cannot go stale)."""
    return hashlib.sha256(open(os.path.join(ROOT, "sufr_amd", "csrc", name), "rb").read()).hexdigest()


best = {}
acc = defaultdict(lambda: [0.0, 0, 0.0])
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "sufr::" not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            key = (short, r["Counter_Name"])
            v = float(r["Counter_Value"]); g = int(r["Grid_Size"])
            a = acc[key]; a[0] += v; a[1] += 1; a[2] = max(a[2], v)
            if key not in best or g > best[key][0]:
                best[key] = (g, v)
with open(out, "w", newline="") as fh:
    fh.write(f"# sufr_part.inc sha256={kernel_source_hash()}\n")
    w = csv.writer(fh)
    w.writerow(["kernel", "counter", "largest_dispatch_grid", "largest_dispatch_value", "mean_value", "dispatches", "max_value"])
    for key in sorted(best):
        g, v = best[key]; s, c, mx = acc[key]
        w.writerow([key[0], key[1], g, v, s / c, c, mx])
print("wrote", out, len(best), "rows")
