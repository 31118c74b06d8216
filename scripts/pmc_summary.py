"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (sum over dispatches)."""
import csv, glob, sys, collections
def short(n):
    for k in ("k_extend", "k_shade", "k_generate", "k_resolve"):
        if k in n:
            return k + ("<count>" if ", true>" in n else "")
    return n[:40]
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        nd = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            nd[k].add(r["Dispatch_Id"])
        print("==", f)
        for k, c in acc.items():
            print(f"{k:22s} dispatches={len(nd[k])}", {n: f"{v:.4g}" for n, v in sorted(c.items())})
