#!/bin/bash
export PSEG_PLAN_FROM_ENV=1   # PSEG_* variables set below become the plan switches of the engines the Python tools create
# ablations on the diagnostic build (results wrong, timing only): tools/ab_diag.sh "<PSEG_DBG=8>" ...
export PSEG_LIB=page-segmentation_amd/csrc/libpseg_diag.so
for v in "" "$@"; do
  r=$(env $v python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], ' '.join('%s=%.1f' % (k[7:] or 'c', v*1e3) for k, v in d['roofline']['per_kernel_ms'].items()))")
  echo "[${v:-default}] $r"
done
