"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace run: tools/kernel_gaps.py <dir> [n]  (prints a window of n launches from the
middle of the trace and the mean gap)"""
import csv, glob, os, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
fs = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = [r for f in fs for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mid = len(rows) // 2
prev = None; gaps = []
for i, r in enumerate(rows):
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev is not None and st - prev < 50000: gaps.append((st - prev) / 1e3)
    if mid <= i < mid + n: print("%-64s dur %7.1f us  gap before %6.2f us" % (r["Kernel_Name"][:64], (en - st) / 1e3, (st - prev) / 1e3 if prev else 0))
    prev = en
print("launches %d  mean gap %.2f us  median %.2f us" % (len(rows), sum(gaps) / max(1, len(gaps)), sorted(gaps)[len(gaps) // 2] if gaps else 0))
