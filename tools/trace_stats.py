"""Per-kernel statistics of a rocprofv3 --kernel-trace run (the sqlite database it writes): python tools/trace_stats.py DB [name filter]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = list(db.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by sum(end-start) desc"))
for r in rows:
    if flt in r[0]:
        print(f"{r[0][:72]:72s} n={r[1]:6d} avg={r[2]/1e3:8.1f} min={r[3]/1e3:7.1f} max={r[4]/1e3:8.1f} us  total={r[5]/1e6:7.2f} ms")
