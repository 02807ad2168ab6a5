#!/bin/bash
# rocprofv3 evidence for one round (run on the GPU box from the repo root):  tools/profile_round.sh r02
#   1. --kernel-trace --stats of the default bench (per-kernel average durations)
#   2. three separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ busy + MFMA busy cycles), no tracing beside them
# Raw output lands in gpurun_out/<tag>_*; the summaries to keep are copied to profiles/ by tools/pmc_traffic.py and by hand.
set -o pipefail
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
CMD="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- $CMD > $OUT/${TAG}_stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${TAG}_fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${TAG}_write.log 2>&1 || exit 3
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/${TAG}_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${TAG}_sq.log 2>&1 || echo "SQ pass failed (counter set not available?)"
find $OUT/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_bf16_kernel_stats.csv \;
python3 tools/pmc_traffic.py $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_traffic.json
echo done
