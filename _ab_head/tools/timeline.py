import csv, sys, glob
f=glob.glob(sys.argv[1])[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t_end=int(rows[-1]['End_Timestamp'])
lo=float(sys.argv[2]); hi=float(sys.argv[3])
sel=[r for r in rows if t_end-lo*1e6 < int(r['Start_Timestamp']) < t_end-hi*1e6]
t0=int(sel[0]['Start_Timestamp'])
for r in sel:
    s=(int(r['Start_Timestamp'])-t0)/1e6; e=(int(r['End_Timestamp'])-t0)/1e6
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:22]
    if e-s>0.1: print("%8.2f %8.2f %7.2f q=%s %s"%(s,e,e-s,r.get('Queue_Id','?'),n))
