"""Top kernels of a rocprofv3 --kernel-trace --stats directory: tools/kstats.py <dir> [n]"""
import csv, glob, os, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i >= n: break
        print("%-78s calls %5s avg %9.1f us total %10.1f us" % (r["Name"][:78], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
