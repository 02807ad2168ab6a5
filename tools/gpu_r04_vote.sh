#!/bin/bash
# round 4: the vote's fused tile pass -- parity tests, then per-kernel durations on configs[4]'s page
set -o pipefail
mkdir -p gpurun_out
export PSEG_PLAN_FROM_ENV=1 TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_post_gpu.py -x -q -m gpu > gpurun_out/vote_tests.log 2>&1 || { tail -40 gpurun_out/vote_tests.log; exit 1; }
tail -3 gpurun_out/vote_tests.log
timeout -k 10 300 python tools/bench_vote.py || exit 1
for v in "" $VARIANTS; do
  rm -rf /tmp/vote_prof
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vote_prof -- python3 tools/bench_vote.py > gpurun_out/vote_prof.log 2>&1 || { tail -20 gpurun_out/vote_prof.log; exit 1; }
  echo "== [$v]"
  find /tmp/vote_prof -name "*kernel_stats.csv" -exec cat {} \; | grep -v rocclr | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin):
    if r[0]=='Name': continue
    print('%-45s calls %s avg %.1f min %.1f max %.1f us' % (r[0].split('(')[0][-45:], r[1], float(r[3])/1e3, float(r[5])/1e3, float(r[6])/1e3))"
done
