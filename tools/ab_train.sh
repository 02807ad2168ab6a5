#!/bin/bash
# same-box A/B of train-step knobs at 2048x1536: tools/ab_train.sh "<ENV=1 ...>" ...   (first variant = default)
for v in "" "$@"; do
  r=$(env $v python tools/bench_train.py --steps 20 --height 2048 --width 1536 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "[${v:-default}] $r ms/step"
done
