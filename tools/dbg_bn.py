"""First layer at which the float32 engine's res_unet + BatchNormalization activations leave the oracle's."""
import sys, os
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")   # PSEG_* of the environment -> plan switches of the engines created here
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "page-segmentation_amd")]
import torch; torch.cuda.is_available()
import pseg_amd as gpu
import oracle
from pseg_amd import synth
oracle.build()
C = 3
Wt = oracle.init_weights("res_unet", C, seed=42, gain=1.5, bias_scale=0.05, batch_norm=True)
img = synth.synth_page(3, 96, 96, C)[0][:64, :96].copy()
z, acts = oracle.models.forward("res_unet", Wt, img, "f32", return_acts=True)
eng = gpu.Engine("res_unet", C, mode=gpu.MODE_F32_EXACT, batch_norm=True)
eng.set_weights(Wt)
zg = eng.predict(img)[0]
for name, a in acts.items():
    if name == "logits":
        continue
    try:
        g = eng.activation(name)
    except Exception as ex:
        print(name, "n/a", ex); continue
    if g.shape != a.shape:
        # BN over a concat runs as two ops: compare the first slice
        a = a[..., :g.shape[-1]]
        if a.shape[:2] != g.shape[:2]:
            a = a[::2, ::2]
    d = np.abs(g - a)
    print("%-28s %-18s maxdiff %.3e  nz %d" % (name, g.shape, d.max(), int((d > 0).sum())))
print("logits", np.abs(zg - z).max())
