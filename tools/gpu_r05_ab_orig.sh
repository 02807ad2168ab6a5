#!/bin/bash
# A/B of the working tree's library against csrc/libpseg_orig.so (a build of the previous commit) on one box
export PSEG_PLAN_FROM_ENV=1
ARCH=${1:-fcn_skip}
STEPS=${2:-30}
run() {
  python bench.py --arch $ARCH --steps $STEPS --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['per_kernel_ms']
print('$1', d['ms_per_step'], d['roofline']['whole_net_frac'], ' '.join('%s=%.1f' % (n[7:] or 'c', v*1e3) for n, v in list(k.items())[:12]))"
}
for r in 1 2 3; do
  PSEG_LIB=page-segmentation_amd/csrc/libpseg_orig.so run "orig" || exit 1
  run "new " || exit 1
done
