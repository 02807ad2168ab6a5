#!/bin/bash
# same-box A/B of float32-engine variants: tools/ab_f32.sh "VAR=val ..." "VAR=val ..."   (each argument = one environment)
set -o pipefail
OUT=$PWD/gpurun_out
for envs in "$@"; do
  echo "== $envs"
  env $envs python3 bench.py --mode f32 --steps 5 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'], json.dumps(d['roofline']['per_kernel_ms']))"
done
