#!/bin/bash
export PSEG_PLAN_FROM_ENV=1   # PSEG_* variables set below become the plan switches of the engines the Python tools create
# rocprofv3 evidence for one round (run on the GPU box from the repo root):  tools/profile_round.sh r03
#   1. --kernel-trace --stats of the default bench (per-kernel average durations of the bf16 page)
#   2. --kernel-trace --stats of every other leg the bench line quotes: float32 engine, train step, unet, res_unet, configs[4]
#   3. separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ busy + MFMA busy cycles), no tracing beside them
# Raw output lands in gpurun_out/<tag>_*; the summaries to keep are copied to profiles/ by tools/pmc_traffic.py and by hand.
set -o pipefail
TAG=${1:-r05}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
stats() {   # stats <name> <program args...>
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${name}_stats -- "$@" > $OUT/${TAG}_${name}_stats.log 2>&1 || return 1
  find $OUT/${TAG}_${name}_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_${name}_kernel_stats.csv \;
}
stats bf16 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra || exit 1
stats f32 python3 bench.py --mode f32 --steps 10 --warmup 3 --no-cpu-baseline --no-extra || exit 2
# (one stream for the trace: with the weight gradients on their second stream two kernels share the chip and a kernel's duration is
# no longer its own cost)
export PSEG_TRAIN_ONE_STREAM=1
stats train python3 tools/bench_train.py --height 2048 --width 1536 --steps 6 --warmup 2 || exit 3
unset PSEG_TRAIN_ONE_STREAM
stats unet python3 bench.py --arch unet --steps 5 --warmup 2 --no-cpu-baseline --no-extra || exit 4
stats res_unet python3 bench.py --arch res_unet --steps 5 --warmup 2 --no-cpu-baseline --no-extra || exit 5
stats config5 python3 tools/bench_config5.py || exit 6          # the uint8 entries bench.py's extra.config5 leg times (--int64: the reference's maps)
stats pages32 python3 bench.py --pages 32 --steps 3 --warmup 1 --no-cpu-baseline --no-extra || exit 6
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_c5_fetch -- python3 tools/bench_config5.py > $OUT/${TAG}_c5_fetch.log 2>&1 || exit 7
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_c5_write -- python3 tools/bench_config5.py > $OUT/${TAG}_c5_write.log 2>&1 || exit 8
python3 tools/pmc_traffic.py $OUT/${TAG}_c5_fetch $OUT/${TAG}_c5_write $OUT/${TAG}_config5_traffic.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${TAG}_fetch.log 2>&1 || exit 7
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${TAG}_write.log 2>&1 || exit 8
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/${TAG}_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${TAG}_sq.log 2>&1 || echo "SQ pass failed (counter set not available?)"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/${TAG}_f32_sq -- python3 bench.py --mode f32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $OUT/${TAG}_f32_sq.log 2>&1 || echo "f32 SQ pass failed"
python3 tools/pmc_traffic.py $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_traffic.json
python3 tools/pmc_sum.py $OUT/${TAG}_sq > $OUT/${TAG}_sq_counters.txt
python3 tools/pmc_sum.py $OUT/${TAG}_f32_sq > $OUT/${TAG}_f32_sq_counters.txt
echo done
