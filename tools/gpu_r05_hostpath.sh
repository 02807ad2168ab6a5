#!/bin/bash
# pseg_predict_batch, pinned host -> pinned host: ramped unit sizes (1, 2, 4 ... at the head and tail of the list) against equal units
export PSEG_PLAN_FROM_ENV=1
for r in 1 2; do for v in ramp noramp; do
# (the A/B switch PSEG_TMP_NO_RAMP existed only while this was measured: both arms now run the ramp)
python - <<'PY'
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "page-segmentation_amd")]
import numpy as np, torch
torch.cuda.is_available()
import pseg_amd, bench
from pseg_amd import synth
eng = pseg_amd.Engine("fcn_skip", 3, mode=pseg_amd.MODE_BF16)
eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
out = []
for n in (32, 8):
    r = bench.leg_host_path(np, pseg_amd, eng, synth, 2048, 1536, 3, n_pages=n, reps=5)
    out.append("%d pages: u8 %.4f i64 %.4f pageable %.4f" % (n, r["uint8"]["ms_per_page"], r["int64"]["ms_per_page"], r["uint8_pageable_via_ring"]["ms_per_page"]))
print("noramp" if os.environ.get("PSEG_TMP_NO_RAMP") else "ramp  ", " | ".join(out))
PY
done; done
