#!/bin/bash
export PSEG_PLAN_FROM_ENV=1   # PSEG_* variables set below become the plan switches of the engines the Python tools create
# round-4 diagnostics pass 1: LDS-DMA ring fill rate vs issuing waves; phase stamps of the 1/8-resolution k5 layers
set -o pipefail
OUT=$PWD/gpurun_out
timeout -k 10 120 tools/microtests/lds_dma_ring.bin > $OUT/r04_dma_ring.txt 2>&1; echo "ring rc=$?"
export PSEG_LIB=page-segmentation_amd/csrc/libpseg_diag.so
timeout -k 10 200 python3 tools/trace_layers.py fcn_skip conv2d_6 conv2d_transpose conv2d_4 conv2d_5 conv2d_transpose_2 > $OUT/r04_trace_default.txt 2>&1; echo "trace rc=$?"
PSEG_DBG=8 timeout -k 10 200 python3 tools/trace_layers.py fcn_skip conv2d_6 conv2d_transpose > $OUT/r04_trace_nowdma.txt 2>&1; echo "trace nowdma rc=$?"
cat $OUT/r04_dma_ring.txt $OUT/r04_trace_*.txt
