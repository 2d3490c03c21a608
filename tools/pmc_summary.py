"""Summarise rocprofv3 --pmc passes (counter_collection.csv) into mean counter value per launch
and kernel.  Usage: pmc_summary.py OUT.json DIR [DIR ...]   (one DIR per --pmc pass)"""
import collections, csv, glob, json, sys
out, dirs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = collections.defaultdict(float)          # a counter may be reported per XCD / SE: sum them
        for r in csv.DictReader(open(f)):
            per_dispatch[(r["Kernel_Name"], r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (kn, cn, _), v in per_dispatch.items():
            a = acc[kn][cn]; a[0] += v; a[1] += 1
res = {kn: {cn: {"launches": a[1], "mean_per_launch": a[0] / a[1]} for cn, a in cs.items()} for kn, cs in sorted(acc.items())}
json.dump(res, open(out, "w"), indent=1)
print("kernels:", len(res))
