"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel table (markdown)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
for n, c, t, a, mn, mx in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    n = n.replace("swn::", "")[:70]
    print(f"| {n} | {c} | {t/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*t/tot:.1f} |")
print(f"\ntotal kernel time {tot/1e6:.2f} ms")
