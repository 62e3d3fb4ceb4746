#!/usr/bin/env python
"""Join tools/pmc_calibrate's known byte counts with rocprofv3's FETCH_SIZE / WRITE_SIZE CSVs.
usage: tools/pmc_calibration_table.py <dir holding fetch/ write/ fetch.log write.log>  ->  the table on stdout"""
import collections, csv, glob, sys

d = sys.argv[1]
known = {}
for line in open(f"{d}/fetch.log"):
    if line.startswith("KNOWN"):
        t = line.split()
        known[t[1]] = (t[2], {kv.split("=")[0]: int(kv.split("=")[1]) for kv in t[3:]})
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write"):
    for f in glob.glob(f"{d}/{sub}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            vals[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("rocprofv3 FETCH_SIZE / WRITE_SIZE (reported in KB) against known byte counts, per launch (second of two launches)")
print(f"{'kernel':18s} {'dir':5s} {'useful MB':>10s} {'64B-line MB':>12s} {'128B-line MB':>13s} {'counter MB':>11s} {'ctr/useful':>10s} {'ctr/64B':>8s} {'ctr/128B':>9s}")
for k, (direction, b) in known.items():
    c = vals.get(k, {}).get("FETCH_SIZE" if direction == "read" else "WRITE_SIZE", [])
    if not c:
        print(f"{k:18s} {direction:5s}  (no counter rows)")
        continue
    mb = c[-1] * 1024 / 1e6
    print(f"{k:18s} {direction:5s} {b['useful']/1e6:10.1f} {b['lines64']/1e6:12.1f} {b['lines128']/1e6:13.1f} {mb:11.1f} {mb*1e6/b['useful']:10.3f} {mb*1e6/b['lines64']:8.3f} {mb*1e6/b['lines128']:9.3f}")
