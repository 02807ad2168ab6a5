#!/bin/bash
# same-box A/B of bf16 knobs with the per-kernel times: tools/ab_knobs.sh <arch> "<ENV=1 ...>" ...
arch=$1; shift
for rep in 1 2; do
  for v in "" "$@"; do
    r=$(env $v python bench.py --arch $arch --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['whole_net_frac'], ' '.join('%s=%.1f' % (k[7:] or 'c', v*1e3) for k, v in d['roofline']['per_kernel_ms'].items()))")
    echo "rep $rep [${v:-default}] $r"
  done
done
