#!/bin/bash
# round 4: vote tile kernel variants on the same box (parity tests first)
set -o pipefail
mkdir -p gpurun_out
export PSEG_PLAN_FROM_ENV=1 TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_post_gpu.py -x -q -m gpu > gpurun_out/vote_tests.log 2>&1 || { tail -40 gpurun_out/vote_tests.log; exit 1; }
tail -1 gpurun_out/vote_tests.log
for v in "" $VARIANTS ""; do
  echo "[$v] $(env $v timeout -k 10 300 python tools/bench_vote.py)"
done
