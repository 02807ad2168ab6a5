#!/bin/bash
export PSEG_PLAN_FROM_ENV=1   # PSEG_* variables set below become the plan switches of the engines the Python tools create
# same-box A/B of two builds of the library: tools/ab_lib.sh <lib_a.so> <lib_b.so> [bench args]
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for L in $A $B; do
    r=$(PSEG_LIB=$L python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['whole_net_frac'], ' '.join('%s=%.1f' % (k[7:] or 'c', v*1e3) for k, v in d['roofline']['per_kernel_ms'].items()))")
    echo "rep $rep [$(basename $L)] $r"
  done
done
