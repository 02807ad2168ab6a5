#!/bin/bash
# round-3 GPU pass A: full GPU test suite, default bench, rocprofv3 kernel stats of configs[4] and the float32 engine
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
TAG=${1:-r03a}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/${TAG}_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/${TAG}_tests.log
tail -5 $OUT/${TAG}_tests.log
timeout -k 10 400 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err && tail -c 600 $OUT/${TAG}_bench.json &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c5 -- python3 tools/bench_config5.py > $OUT/${TAG}_c5.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_f32 -- python3 bench.py --mode f32 --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $OUT/${TAG}_f32.log 2>&1
echo "pass A rc=$?"
find $OUT/${TAG}_c5 -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_c5_kernel_stats.csv \;
find $OUT/${TAG}_f32 -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_f32_kernel_stats.csv \;
