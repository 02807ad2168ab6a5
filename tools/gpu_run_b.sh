#!/bin/bash
# GPU pass B: full GPU test suite + float32 engine timing (per-layer) + train step timing
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
TAG=${1:-r03b}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/${TAG}_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/${TAG}_tests.log
tail -15 $OUT/${TAG}_tests.log
timeout -k 10 300 python3 bench.py --mode f32 --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $OUT/${TAG}_f32.json 2> $OUT/${TAG}_f32.err; echo "f32 rc=$?"
python3 -c "
import json;d=json.load(open('$OUT/${TAG}_f32.json'));print(d['ms_per_step'], json.dumps(d['roofline']['per_kernel_ms']))"
timeout -k 10 300 python3 tools/bench_train.py --height 2048 --width 1536 --steps 6 --warmup 2 > $OUT/${TAG}_train.json 2> $OUT/${TAG}_train.err; echo "train rc=$?"; cut -c1-300 $OUT/${TAG}_train.json
