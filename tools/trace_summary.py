"""Summarise gpurun_out/trace_<layer>.bin (PSEG_TRACE diagnostic: per-workgroup s_memtime stamps
0 start, 1 input tile staged (issued), 2 first barrier passed, 3 k-loop done, 4 epilogue done,
5 HW_ID)."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 12)
t = a[:, :7].astype(np.int64)
ok = t[:, 6] > 0
t = t[ok]
t0 = t[:, 0].min()
d = np.diff(t, axis=1)
print("workgroups", len(t))
for i, n in enumerate(["ring prologue issue", "table copy", "tile DMA issue", "wait + first barrier", "k-loop", "epilogue"]):
    print("%-24s median %8d  p10 %8d  p90 %8d" % (n, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
print("%-24s median %8d" % ("total per WG", np.median(t[:, 6] - t[:, 0])))
