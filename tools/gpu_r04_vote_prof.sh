#!/bin/bash
# round 4: per-kernel durations of the vote on configs[4]'s page (tools/bench_vote.py under rocprofv3 --kernel-trace --stats)
set -o pipefail
mkdir -p gpurun_out
export PSEG_PLAN_FROM_ENV=1 TMPDIR=/tmp
rm -rf gpurun_out/vote_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vote_prof -- python3 tools/bench_vote.py > gpurun_out/vote_prof.log 2>&1 || { tail -20 gpurun_out/vote_prof.log; exit 1; }
find gpurun_out/vote_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/vote_kernel_stats.csv \;
cut -c1-160 gpurun_out/vote_kernel_stats.csv
