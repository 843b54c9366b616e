"""Dev tool: export per-kernel statistics and one step's timeline from a rocprofv3 rocpd database (results.db).
usage: python scripts/rocpd_export.py <results.db> <out_prefix>   -> <out_prefix>_kernel_stats.csv, _step_timeline.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = [dict(zip(cols, r)) for r in cur.execute("select * from kernels")]


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]


ks = sorted((r["start"], r["end"], short(r["name"]), r.get("queue_id", "")) for r in rows)
stats = {}
for s, e, n, q in ks:
    st = stats.setdefault(n, [0, 0, 1 << 62, 0])
    d = e - s
    st[0] += 1; st[1] += d; st[2] = min(st[2], d); st[3] = max(st[3], d)
tot = sum(v[1] for v in stats.values())
with open(sys.argv[2] + "_kernel_stats.csv", "w") as fh:
    fh.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n")
    for n, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        fh.write('"%s",%d,%d,%.1f,%d,%d,%.2f\n' % (n, v[0], v[1], v[1] / v[0], v[2], v[3], 100.0 * v[1] / tot))
idx = [i for i, k in enumerate(ks) if "begin_step" in k[2]]
if len(idx) > 4:
    i0, i1 = idx[-3], idx[-2]
    t0 = ks[i0][0]
    with open(sys.argv[2] + "_step_timeline.txt", "w") as fh:
        fh.write("# one step (third from last) : start us, end us, duration us, queue, kernel\n")
        for s, e, n, q in ks[i0:i1 + 1]:
            fh.write("%8.1f %8.1f %7.1f  q%s %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
