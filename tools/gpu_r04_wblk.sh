#!/bin/bash
# round 4: channel-blocked flat weight gradients (unet / res_unet) -- gradient tests, then train-step timing with / without
set -o pipefail
mkdir -p gpurun_out
export PSEG_PLAN_FROM_ENV=1 TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests/test_train_arch_gpu.py tests/test_batchnorm_gpu.py -x -q -m gpu > gpurun_out/wblk_tests.log 2>&1 || { tail -40 gpurun_out/wblk_tests.log; exit 1; }
tail -3 gpurun_out/wblk_tests.log
for arch in unet res_unet; do
  for v in "" "PSEG_WGRAD_NO_BLK=1"; do
    r=$(env $v timeout -k 10 300 python tools/bench_train.py --arch $arch --height 512 --width 384 --steps 8 --warmup 3 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],2))")
    echo "$arch [$v] $r"
  done
done
bash tools/prof_train_arch.sh unet 512 384 > gpurun_out/unet_tprof.txt 2>&1; head -14 gpurun_out/unet_tprof.txt | cut -c1-140
