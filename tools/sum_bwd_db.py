import sqlite3,sys,glob
for d in sys.argv[1:]:
    db=sqlite3.connect(glob.glob(d+"/**/*.db",recursive=True)[0])
    tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd=[t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks=[t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q="select s.kernel_name, d.grid_size_x, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from %s d join %s s on d.kernel_id=s.id where s.kernel_name like '%%costvol_bwd%%' group by s.kernel_name, d.grid_size_x order by 1,2"%(kd,ks)
    print(d)
    for r in db.execute(q): print("   %-70s grid %8d n=%3d avg %8.1f min %8.1f"%(r[0][:70],r[1],r[2],r[3],r[4]))
