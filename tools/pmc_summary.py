#!/usr/bin/env python
"""Summarise a tools/pmc.sh output directory: per-kernel averages of each counter."""
import collections, csv, glob, sys
d = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "k_iter"
for f in sorted(glob.glob(f"{d}/*/*/*_kernel_stats.csv")):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("STATS", r["Name"][:70], r["Calls"], "avg_us=%.1f" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
for f in sorted(glob.glob(f"{d}/*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"])
    for k, cs in agg.items():
        for c, v in cs.items():
            print(f"{k:60s} {c:32s} n={len(v):3d} avg={sum(v)/len(v):.6g} last={v[-1]:.6g}")
