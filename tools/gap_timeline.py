#!/usr/bin/env python
"""dev: idle gaps between consecutive kernels of the LAST run in a rocprofv3 kernel trace (csv): per kernel its duration and the gap before it.
usage: gap_timeline.py <kernel_trace.csv> [n_last]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 70
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
prev_end = None
tot_gap = tot_k = 0.0
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{r['Kernel_Name'].split('(')[0][:58]:60s} dur {((en - st) / 1e3):8.1f} us   gap before {gap:8.1f} us")
    if prev_end is not None:
        tot_gap += max(gap, 0.0)
    tot_k += (en - st) / 1e3
    prev_end = en
print(f"total kernel time {tot_k:.1f} us, total gaps {tot_gap:.1f} us over {len(rows)} kernels")
