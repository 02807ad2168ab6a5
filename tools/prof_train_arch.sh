#!/bin/bash
# tools/prof_train_arch.sh <arch> <H> <W> : rocprofv3 kernel stats of the train step of another graph
ARCH=${1:-unet}; H=${2:-512}; W=${3:-384}
export TMPDIR=/tmp
python3 tools/bench_train.py --arch $ARCH --height $H --width $W --steps 6 --warmup 2 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/${ARCH}_tprof -- python3 tools/bench_train.py --arch $ARCH --height $H --width $W --steps 6 --warmup 2 > /dev/null 2>&1 || exit 2
find $PWD/gpurun_out/${ARCH}_tprof -name "*kernel_stats.csv" -exec cp {} $PWD/gpurun_out/${ARCH}_train_kernel_stats.csv \;
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/${ARCH}_train_kernel_stats.csv')))
print(sum(int(r['TotalDurationNs']) for r in rows)/8/1e6)
for r in rows[:22]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls'])/8:5.1f} {float(r['AverageNs'])/1e3:9.1f} {int(r['TotalDurationNs'])/8/1e3:9.1f}")
PY
