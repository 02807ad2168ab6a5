#!/bin/bash
# round 4: the vote's tile pass -- parity tests, configs[4] leg, per-kernel stats of tools/bench_config5.py
set -o pipefail
mkdir -p gpurun_out
export PSEG_PLAN_FROM_ENV=1 TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_post_gpu.py tests/test_configs_gpu.py -x -q -m gpu > gpurun_out/vote_tests.log 2>&1 || { tail -40 gpurun_out/vote_tests.log; exit 1; }
tail -3 gpurun_out/vote_tests.log
timeout -k 10 300 python tools/bench_vote.py || exit 1
timeout -k 10 300 python tools/bench_config5.py || exit 1
rm -rf gpurun_out/r04_config5_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_config5_stats -- python3 tools/bench_config5.py > gpurun_out/r04_config5_stats.log 2>&1 || exit 1
find gpurun_out/r04_config5_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_config5_kernel_stats.csv \;
grep -E "vote|ccl" gpurun_out/r04_config5_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,150-260
