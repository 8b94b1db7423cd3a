import sqlite3,sys
db=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
q=f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3*count(*) desc"
for r in db.execute(q): print(f"{r[0][:70]:70s} n={r[1]:5d} avg={r[2]:9.1f}us min={r[3]:9.1f}")
