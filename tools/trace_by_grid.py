"""Aggregate a rocprofv3 kernel trace by (kernel, grid): launches per step, average duration, a 10-us histogram.
python tools/trace_by_grid.py <t_kernel_trace.csv> <steps in the trace> [rows]"""
import csv, collections, re, sys
f=sys.argv[1]; steps=float(sys.argv[2]) if len(sys.argv)>2 else 1.0
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    short=n.replace('(anonymous namespace)::',''); short=re.sub(r'^void ','',short); short=re.sub(r'\(.*','',short)
    short=re.sub(r'at::native::','',short)[:70]
    key=(short, r['Grid_Size_X'],r['Grid_Size_Y'],r['Grid_Size_Z'],r['Workgroup_Size_X'])
    agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
tot=sum(sum(v) for v in agg.values())
print("total %.2f ms per step"%(tot/1e3/steps))
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv)>3 else 60]:
    wgs=int(k[1])//int(k[4])*int(k[2])*int(k[3])
    import collections as C
    b=C.Counter(int(round(x/10.0))*10 for x in v)
    print("%-70s grid %s,%s,%s wgs %6d  n/step %5.1f avg %7.1f us  per step %6.3f ms | %s"%(k[0], int(k[1])//int(k[4]),k[2],k[3], wgs, len(v)/steps, sum(v)/len(v), sum(v)/1e3/steps, " ".join("%d:%d"%(d,c) for d,c in sorted(b.items()))))
