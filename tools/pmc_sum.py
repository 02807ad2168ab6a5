"""Average per launch of every counter in a rocprofv3 --pmc CSV directory, per kernel: tools/pmc_sum.py <dir> [substring]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "pseg::"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(fn) as f:
        for r in csv.DictReader(f):
            if sub in r["Kernel_Name"]:
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k[:100])
    for c, v in sorted(cs.items()):
        print("    %-32s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
