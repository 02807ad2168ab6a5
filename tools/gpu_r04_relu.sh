#!/bin/bash
# round 4: pre-activation ReLU on the fragments (res_unet) -- parity tests, then res_unet / unet bf16 legs against the previous build
set -o pipefail
mkdir -p gpurun_out
export PSEG_PLAN_FROM_ENV=1
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "res_unet or unet or arch or variants" > gpurun_out/relu_tests.log 2>&1 || { tail -30 gpurun_out/relu_tests.log; exit 1; }
tail -2 gpurun_out/relu_tests.log
for lib in "" page-segmentation_amd/csrc/libpseg_old.so "" page-segmentation_amd/csrc/libpseg_old.so; do
  r=$(PSEG_LIB=$lib timeout -k 10 300 python bench.py --arch res_unet --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "res_unet [${lib:-new}] $r"
done
