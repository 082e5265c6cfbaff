import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
key=sys.argv[2]
idx=[i for i,r in enumerate(rows) if key in r['Kernel_Name']]
a,b=idx[-3],idx[-2]
t0=int(rows[a]['Start_Timestamp']); prev=None; gaps=0; busy=0
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(s-prev)/1e3 if prev else 0
    if prev and s>prev: gaps+=gap
    busy+=(e-s)/1e3
    name=r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','').replace('rocprim::ROCPRIM_400200_NS::detail::','rp::').replace('rocprim::ROCPRIM_400001_NS::detail::','rp1::')
    print("%8.1f +%6.1f gap %6.1f  %s"%((s-t0)/1e3,(e-s)/1e3,gap,name[:70]))
    prev=max(e,prev or 0)
print("step span", (int(rows[b]['Start_Timestamp'])-t0)/1e3, "busy", busy, "gaps", gaps)
