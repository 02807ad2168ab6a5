#!/bin/bash
# GPU pass C: float32 engine timing + SQ counters (MFMA busy, LDS) per kernel; train-arch tests
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
TAG=${1:-r03c}
timeout -k 10 600 python3 -m pytest tests/test_train_arch_gpu.py tests/test_predict_gpu.py -x -q > $OUT/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/${TAG}_tests.log
timeout -k 10 300 python3 bench.py --mode f32 --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $OUT/${TAG}_f32.json 2> $OUT/${TAG}_f32.err; echo "f32 rc=$?"
python3 -c "
import json;d=json.load(open('$OUT/${TAG}_f32.json'));print(d['ms_per_step'], json.dumps(d['roofline']['per_kernel_ms']))"
rocprofv3 --list-avail > $OUT/${TAG}_avail.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/${TAG}_sq1 -- python3 bench.py --mode f32 --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $OUT/${TAG}_sq1.log 2>&1 && python3 tools/pmc_sum.py $OUT/${TAG}_sq1 > $OUT/${TAG}_sq1.txt
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/${TAG}_sq2 -- python3 bench.py --mode f32 --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $OUT/${TAG}_sq2.log 2>&1 && python3 tools/pmc_sum.py $OUT/${TAG}_sq2 > $OUT/${TAG}_sq2.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/${TAG}_sq3 -- python3 bench.py --mode f32 --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $OUT/${TAG}_sq3.log 2>&1 && python3 tools/pmc_sum.py $OUT/${TAG}_sq3 > $OUT/${TAG}_sq3.txt
echo "pass C done"
