#!/bin/bash
export PSEG_PLAN_FROM_ENV=1   # PSEG_* variables set below become the plan switches of the engines the Python tools create
# conv_sp_kernel: phase stamps, default and with the loaders' DMAs switched off (diagnostic build, wrong results)
OUT=$PWD/gpurun_out
timeout -k 10 200 python3 tools/sp_trace.py > $OUT/r04_sp_trace_rel.txt 2>&1
export PSEG_LIB=page-segmentation_amd/csrc/libpseg_diag.so
for d in 0 1 2 3; do
  echo "== PSEG_SP_DBG=$d" >> $OUT/r04_sp_trace_ab.txt
  PSEG_SP_DBG=$d timeout -k 10 200 python3 tools/sp_trace.py >> $OUT/r04_sp_trace_ab.txt 2>&1
done
cat $OUT/r04_sp_trace_rel.txt $OUT/r04_sp_trace_ab.txt | grep -v amdgpu.ids
