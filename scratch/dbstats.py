"""Per-kernel totals from a rocprofv3 rocpd database: python scratch/dbstats.py results.db [steps]"""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms  ({tot/1e6/steps:.2f} ms/step)")
for n, k, t, a in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    n = re.sub(r"\(.*", "", n)
    print(f"{n[:80]:80s} n={k:5d} ms/step={t/1e6/steps:7.3f} avg_us={a/1e3:8.1f} {100*t/tot:5.1f}%")
