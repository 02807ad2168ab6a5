#!/bin/bash
# round 4: fused upsample -> k2 conv (unet) -- parity tests, then the unet leg with / without (PSEG_UPSPLIT_TWO_PASS)
set -o pipefail
mkdir -p gpurun_out
export PSEG_PLAN_FROM_ENV=1
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py tests/test_configs_gpu.py tests/test_predict_gpu.py -x -q -m gpu -k "unet or arch or variants or upsplit" > gpurun_out/upf_tests.log 2>&1 || { tail -30 gpurun_out/upf_tests.log; exit 1; }
tail -2 gpurun_out/upf_tests.log
for v in "" "PSEG_UPSPLIT_LDS=1" "PSEG_UPSPLIT_TWO_PASS=1" "" "PSEG_UPSPLIT_LDS=1"; do
  r=$(env $v timeout -k 10 300 python bench.py --arch unet --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pk=d['roofline']['per_kernel_ms']; print(d['ms_per_step'], ' '.join('%s=%.0f'%(k[7:],pk[k]*1e3) for k in ('conv2d_10','conv2d_13','conv2d_16','conv2d_19')))")
  echo "unet [${v:-fused}] $r"
done
