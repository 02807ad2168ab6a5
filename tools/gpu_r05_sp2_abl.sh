#!/bin/bash
# ablations of conv_sp2_kernel's k-loop (diagnostic build, wrong results): which resource bounds it
export PSEG_PLAN_FROM_ENV=1 PSEG_LIB=page-segmentation_amd/csrc/libpseg_diag.so
for d in ${@:-0 16 3 96 99 115}; do
  echo "== PSEG_SP_DBG=$d (1 no weight DMA, 2 no tile DMA, 16 no dynamic priority, 32 no pixel fragment reads, 64 no weight fragment reads)"
  PSEG_SP_DBG=$d python tools/sp2_trace.py conv2d_5 conv2d_transpose_2 2>&1 | grep -v amdgpu.ids | cut -c1-330
done
