#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 run from its rocpd sqlite database (ROCm 7 default output).  Usage: prof_db.py results.db [N]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"{'kernel':72s} {'calls':>7s} {'total ms':>10s} {'avg us':>9s} {'min us':>8s} {'max us':>9s} {'%':>6s}")
for r in rows[:top]:
    print(f"{r[0][:72]:72s} {r[1]:7d} {r[2]/1e6:10.3f} {r[3]/1e3:9.2f} {r[4]/1e3:8.2f} {r[5]/1e3:9.2f} {100*r[2]/tot:6.1f}")
print(f"{'all kernels':72s} {sum(r[1] for r in rows):7d} {tot/1e6:10.3f}")
