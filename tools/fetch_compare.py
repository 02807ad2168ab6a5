"""Sum FETCH_SIZE (KiB, rocprofv3 --pmc FETCH_SIZE csv) per kernel for two result directories: HBM read traffic with / without a change.
    python tools/fetch_compare.py gpurun_out/dirA gpurun_out/dirB"""
import csv, glob, os, sys, collections
def load(d):
    rows = collections.defaultdict(list)
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if r.get("Counter_Name") == "FETCH_SIZE" and "pseg::" in r["Kernel_Name"]:
                rows[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    return rows
a, b = load(sys.argv[1]), load(sys.argv[2])
for k in sorted(set(a) | set(b)):
    fa = sum(a.get(k, [0])) / max(len(a.get(k, [])), 1) * 2048 / 1e6
    fb = sum(b.get(k, [0])) / max(len(b.get(k, [])), 1) * 2048 / 1e6
    print("%-72s %8.1f MB  %8.1f MB" % (k, fa, fb))
