"""Probe (not a test): loss trajectories of the unet / res_unet train step over three identical runs -- how much the float atomics move them."""
import sys
sys.path[:0] = [".", "page-segmentation_amd"]
import numpy as np, torch
torch.cuda.is_available()
import pseg_amd as gpu
from pseg_amd import synth
import oracle as O; O.build()
def sample(seed):
    img, _, mask = synth.synth_page(seed, 96, 96, 3)
    return np.ascontiguousarray(img[:64,:64]), np.ascontiguousarray(mask[:64,:64])
for arch in ("res_unet", "unet"):
  for trial in range(3):
    Wt = O.init_weights(arch, 3, seed=5, gain=1.0, bias_scale=0.02)
    pages = [sample(s) for s in (0, 1)]
    eng = gpu.Engine(arch, 3, mode=gpu.MODE_F32_EXACT)
    eng.set_weights(Wt); eng.train_init(clipnorm=1.0)
    first = np.mean([eng.eval_step(*p)[0] for p in pages])
    tr = []
    for step in range(60):
        eng.train_forward_backward(*pages[step % 2]); eng.train_apply(1e-3)
        if step % 10 == 9: tr.append(round(float(np.mean([eng.eval_step(*p)[0] for p in pages])), 4))
    print(arch, trial, round(float(first), 4), tr)
    eng.close()
