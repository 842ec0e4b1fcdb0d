import sqlite3,sys
db=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
q=f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc limit 16"
for r in db.execute(q): print("%-100s %6d %9.1f us  %10.1f ms"%(r[0][:100],r[1],r[2]/1e3,r[3]/1e6))
# last replay: gaps
rows=list(db.execute(f"select d.start,d.end,s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
n=len(rows); tail=rows[-560:]
busy=sum(e-s for s,e,_ in tail); span=tail[-1][1]-tail[0][0]
print("last 560 launches: span %.1f us busy %.1f us"%(span/1e3,busy/1e3))
