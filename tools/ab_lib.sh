#!/bin/bash
# default bench (bf16 page, no extra legs) on several builds of the library, alternating: tools/ab_lib.sh <lib.so> <lib.so> ...
export PSEG_PLAN_FROM_ENV=1
for rep in 1 2; do
for lib in "$@"; do
  r=$(PSEG_LIB=$lib python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], ' '.join('%s=%.1f' % (k[7:] or 'c', v*1e3) for k, v in d['roofline']['per_kernel_ms'].items()))")
  echo "[$lib] $r"
done
done
