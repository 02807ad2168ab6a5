#!/bin/bash
# round 4: SQ counters of the vote kernels (tools/bench_vote.py), one rocprofv3 --pmc pass per counter group
set -o pipefail
mkdir -p gpurun_out
export PSEG_PLAN_FROM_ENV=1 TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_LDS_ATOMIC_RETURN SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/vpmc$i
  rocprofv3 --pmc $grp --output-format csv -d /tmp/vpmc$i -- python3 tools/bench_vote.py > gpurun_out/vote_pmc$i.log 2>&1 || { tail -5 gpurun_out/vote_pmc$i.log; continue; }
  python3 tools/pmc_sum.py /tmp/vpmc$i vote_tile
done
