import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'vertex_normal_k' in r['Kernel_Name']]
i0=idx[len(idx)//2]; i1=idx[len(idx)//2+1]
t0=int(rows[i0]['Start_Timestamp']); prev_end=t0
for r in rows[i0:i1]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print("%-44s start %7.1f dur %6.1f gap %5.1f"%(r['Kernel_Name'][:44].replace('gs::',''), (s-t0)/1e3,(e-s)/1e3,(s-prev_end)/1e3))
    prev_end=e
print("step total us", (int(rows[i1]['Start_Timestamp'])-t0)/1e3)
