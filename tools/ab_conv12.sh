#!/bin/bash
export PSEG_PLAN_FROM_ENV=1   # PSEG_* variables set below become the plan switches of the engines the Python tools create
# A/B of conv12_ws_kernel variants inside ONE gpurun call (same box, same clocks): prints conv2d_1 us per variant, 3 rounds
for r in 1 2 3; do
for v in "" "PSEG_WS_FORM=2" "PSEG_WS_FORM=1" "PSEG_NO_C32=1" "PSEG_NO_WS=1"; do
  env $v python bench.py --no-cpu-baseline --no-extra --steps 30 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%-34s conv2d_1 %.2f us  page %.4f ms' % ('$v' or 'default', d['roofline']['per_kernel_ms']['conv2d_1']*1e3, d['ms_per_step']))"
done; done
