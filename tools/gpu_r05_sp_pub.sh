#!/bin/bash
# conv_sp_kernel's weight loader publishing landed groups while the ring is full: tests + the page's per-kernel times
export PSEG_PLAN_FROM_ENV=1
python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "streamed or two_team" 2>&1 | tail -2
for r in 1 2 3; do
python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['per_kernel_ms']
print(d['ms_per_step'], ' '.join('%s=%.1f' % (n[7:] or 'c', v*1e3) for n, v in k.items()))"
done
