#!/bin/bash
# A/B of conv_sp2_kernel on the headline page: alternating runs in one box (per-kernel us of the quarter-resolution trio)
export PSEG_PLAN_FROM_ENV=1
run() {
  python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['per_kernel_ms']
print('$1', d['ms_per_step'], ' '.join('%s=%.1f' % (n[7:] or 'c', v*1e3) for n, v in k.items()), 'trio=%.1f' % ((k['conv2d_4']+k['conv2d_5']+k['conv2d_transpose_2'])*1e3))"
}
for r in 1 2; do
  run "default     " || exit 1
  PSEG_SP2=0 run "PSEG_SP2=0  " || exit 1
  PSEG_SP2=32 run "PSEG_SP2=32 " || exit 1
done
