"""per-kernel statistics out of a rocprofv3 (rocpd sqlite) kernel trace: python tools/kstats.py <results.db> [name filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
q = "select name, count(*), avg(end-start)/1000.0, min(end-start)/1000.0, max(end-start)/1000.0, sum(end-start)/1000.0 from kernels group by name order by 6 desc"
print("%-100s %6s %10s %10s %10s %12s" % ("kernel", "calls", "avg us", "min us", "max us", "total us"))
for r in c.execute(q):
    if flt in r[0]:
        print("%-100s %6d %10.1f %10.1f %10.1f %12.1f" % (r[0][:100], r[1], r[2], r[3], r[4], r[5]))
