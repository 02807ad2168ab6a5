#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of the headline page under the plan switches given in the environment
export PSEG_PLAN_FROM_ENV=1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
name=${1:-r05_ks}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $OUT/${name}.log 2>&1 || exit 1
find $OUT/${name}_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \;
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/${name}_kernel_stats.csv")))
for r in rows[:14]:
    print("%-110s calls %5s avg %8.1f us" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3))
PY
