"""Per-kernel PMC counter values (mean over dispatches, summed over dimensions) from a rocprofv3
results .db (rocpd sqlite). Usage: dbpmc.py results.db [kernel-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "fused"
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
def tab(prefix):
    return [t for t in tabs if t.startswith(prefix)][0]
kd, ks, pc, pi = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
cols = [r[1] for r in db.execute(f"pragma table_info({pc})")]
q = (f"select p.name, d.id, sum(e.value) from {pc} e join {pi} p on e.pmc_id = p.id "
     f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
     f"where s.kernel_name like '%{pat}%' group by p.name, d.id")
acc = {}
for name, did, val in db.execute(q):
    acc.setdefault(name, []).append(val)
for name, vals in sorted(acc.items()):
    print(f"{name:36s} n={len(vals):3d} mean={sum(vals) / len(vals):16.1f}")
