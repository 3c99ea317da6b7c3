"""Sum rocprofv3 --pmc counter_collection.csv per kernel name and counter (per-dispatch average).  Usage: pmc_sum.py DIR [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "advance"
acc = collections.defaultdict(lambda: [0.0, set()])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1].add(r["Dispatch_Id"])
for (kn, cn), (v, ids) in sorted(acc.items()):
    print(f"{kn:60s} {cn:28s} {v / max(1, len(ids)):16.1f}  ({len(ids)} dispatches)")
