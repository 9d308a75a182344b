import sqlite3, glob, sys
for d in sys.argv[1:]:
    f = glob.glob(d+"/**/*.db", recursive=True)[0]
    c = sqlite3.connect(f)
    tabs=[r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kt=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
    q=f"select s.kernel_name, count(*), avg(k.end-k.start), sum(k.end-k.start) from {kt} k join {ks} s on k.kernel_id=s.id group by s.kernel_name order by 1"
    print("==",d); tot=0
    for n,cnt,avg,sm in c.execute(q):
        if 'frt' not in n: continue
        n=n.replace("_ZN3frt","").split("ENS_")[0]; tot+=sm
        print(f"  {n:34s} {cnt:4d} avg_us {avg/1e3:8.1f}")
    print("  kernels per frame ms", tot/24/1e6)
