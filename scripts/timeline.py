"""Dev probe: print the kernel timeline of one step from a rocprofv3 kernel_trace.csv."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:40]
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '')) for r in rows)
idx = [i for i, k in enumerate(ks) if 'begin_step' in k[2]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
i0, i1 = idx[which], idx[which + 1]
t0 = ks[i0][0]
for s, e, n, q in ks[i0:i1 + 1]:
    print(f"{(s-t0)/1e3:8.1f} {(e-t0)/1e3:8.1f} {(e-s)/1e3:7.1f}  q{q} {n}")
