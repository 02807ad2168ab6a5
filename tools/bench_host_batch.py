"""pseg_predict_batch from pinned host memory to pinned host memory (SURVEY 8d's boundary metric), 8 pages of 2048x1536."""
import os, sys
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")   # PSEG_* of the environment -> plan switches of the engines created here
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "page-segmentation_amd")]
import numpy as np, torch
torch.cuda.is_available()
import pseg_amd, bench
from pseg_amd import synth
eng = pseg_amd.Engine("fcn_skip", 3, mode=pseg_amd.MODE_BF16)
eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
r = bench.leg_host_path(np, pseg_amd, eng, synth, 2048, 1536, 3, n_pages=8, reps=5)
print({k: v for k, v in os.environ.items() if k.startswith("PSEG_")}, r["uint8"], r["int64"], r["uint8_pageable_via_ring"])
