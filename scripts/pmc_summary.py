"""Aggregate rocprofv3 --pmc counter_collection CSVs: per kernel name and counter, dispatch count and mean value."""
import csv, sys, glob, collections, json

def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]

def summarise(path):
    acc = collections.defaultdict(lambda: [0, 0.0, ""])
    for r in csv.DictReader(open(path)):
        k = (short(r["Kernel_Name"]), r["Counter_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
        acc[k][2] = r.get("Grid_Size", "")
    return {k: (n, s / n, g) for k, (n, s, g) in acc.items()}

if __name__ == "__main__":
    out = {}
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/*/*counter_collection.csv"):
            for (kern, ctr), (n, mean, grid) in sorted(summarise(f).items()):
                if "at::" in kern or "rocprim" in kern or "rocclr" in kern:
                    continue
                out.setdefault(kern, {})[ctr] = {"dispatches": n, "mean": mean}
    print(json.dumps(out, indent=1))
