"""dev: the first lines of a rocprofv3 kernel-stats csv (name, calls, average us, total ms).  usage: kstats_top.py <dir> [n]"""
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv") + glob.glob(sys.argv[1] + "/*_kernel_stats.csv")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in list(csv.DictReader(open(fs[0])))[:n]:
    print(f"{r['Name'][:84]:84s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs']) / 1e3:9.1f} total_ms {float(r['TotalDurationNs']) / 1e6:8.2f} min_us {float(r['MinNs']) / 1e3:8.1f} max_us {float(r['MaxNs']) / 1e3:8.1f}")
