#!/bin/bash
# page units of the 3x3 graphs: tests, then ms per page of 8 pages as one unit against page by page
export PSEG_PLAN_FROM_ENV=1
python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "page_units" 2>&1 | tail -3 || exit 1
for arch in unet res_unet; do
  for mode in "" "--page-by-page"; do
    python bench.py --arch $arch --pages 8 --steps 3 --warmup 1 --no-extra --no-cpu-baseline $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$arch', '$mode' or 'unit of 8', 'ms per page', round(d['ms_per_step']/8, 4), 'whole_net_frac', d['roofline']['whole_net_frac'])"
  done
done
