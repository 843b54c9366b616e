"""FETCH_SIZE / WRITE_SIZE of the two gemm_kk launches of a step separately (they alternate in dispatch order: output
layer dW = A operand 16-byte loads of the B x G d-logits; first layer dW = B operand 8-byte loads of the B x G input)."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if "gemm_kk_kernel" in r["Kernel_Name"]]
        byc = collections.defaultdict(list)
        for r in rows:
            byc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for c, v in byc.items():
            v.sort()
            even, odd = [x for i, (_, x) in enumerate(v) if i % 2 == 0], [x for i, (_, x) in enumerate(v) if i % 2 == 1]
            print(c, "first-in-step launches: n=%d mean=%.0f | second: n=%d mean=%.0f" % (len(even), sum(even) / len(even), len(odd), sum(odd) / max(1, len(odd))))
