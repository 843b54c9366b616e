"""Dev probe: top kernels of a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}% calls={r['Calls']:>6} avg={float(r['AverageNs'])/1e3:8.1f}us {n}")
