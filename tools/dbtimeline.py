"""Print the kernel timeline (start offset, duration, gap to previous kernel) of one LM round from a
rocprofv3 results .db (rocpd sqlite). Usage: dbtimeline.py results.db [first_dispatch] [count]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
count = int(sys.argv[3]) if len(sys.argv) > 3 else 12
prev = None
for name, s, e in rows[first:first + count]:
    gap = (s - prev) / 1000.0 if prev is not None else 0.0
    short = name.split("calib")[-1][:40]
    print(f"{short:42s} dur {(e - s) / 1000.0:8.1f} us   gap {gap:6.1f} us")
    prev = e
