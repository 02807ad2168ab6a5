#!/bin/bash
# round 4: conv_sp_kernel on tiles of 8 x 24 -- bit identity tests, then an A/B of the default bench with / without (PSEG_NO_SP24)
mkdir -p gpurun_out
set -o pipefail
export PSEG_PLAN_FROM_ENV=1
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu -k "streamed or default_plan or page_units" > gpurun_out/tw24_tests.log 2>&1 || { tail -30 gpurun_out/tw24_tests.log; exit 1; }
tail -3 gpurun_out/tw24_tests.log
for v in "" "PSEG_NO_SP24=1" "" "PSEG_NO_SP24=1"; do
  r=$(env $v timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], ' '.join('%s=%.1f' % (k[7:] or 'c', v*1e3) for k, v in d['roofline']['per_kernel_ms'].items()))")
  echo "[${v:-default}] $r"
done
