#!/bin/bash
export PSEG_PLAN_FROM_ENV=1
python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "two_team" > gpurun_out/r05_t4.log 2>&1 || { tail -30 gpurun_out/r05_t4.log; exit 1; }
tail -2 gpurun_out/r05_t4.log
export PSEG_LIB=page-segmentation_amd/csrc/libpseg_diag.so
echo "== dynamic priority"; python tools/sp2_trace.py 2>&1 | grep -v amdgpu.ids
echo "== PSEG_SP_DBG=16 (no dynamic priority)"; PSEG_SP_DBG=16 python tools/sp2_trace.py 2>&1 | grep -v amdgpu.ids
unset PSEG_LIB
bash tools/gpu_r05_sp2_ab.sh
