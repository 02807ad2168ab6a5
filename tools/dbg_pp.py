"""conv_pp_kernel (ping-pong persistent mid-layer kernel) against the three-workgroup instances: same bits?"""
import sys, os
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")   # PSEG_* of the environment -> plan switches of the engines created here
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "page-segmentation_amd")]
import torch; torch.cuda.is_available()
import pseg_amd as gpu
from pseg_amd import synth
for arch in ("fcn_skip", "fcn"):
    for (H, W) in ((2048, 1536), (1024, 768), (1100, 1300)):
        img = synth.synth_page(5, H, W, 3)[0]
        outs = []
        for knob in (None, "PSEG_NO_PP"):
            if knob: os.environ[knob] = "1"
            e = gpu.Engine(arch, 3, mode=gpu.MODE_BF16)
            e.set_weights(synth.glorot_weights(e.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
            z, _, l = e.predict(img, want_probs=False)
            a3 = e.activation("conv2d_2")
            try: p4 = e.activation("max_pooling2d_1")
            except Exception as ex: p4 = None
            outs.append((z, l, a3, p4))
            e.close()
            if knob: os.environ.pop(knob)
        print(arch, H, W, "logits equal", np.array_equal(outs[0][0], outs[1][0]), "conv3 equal", np.array_equal(outs[0][2], outs[1][2]),
              "pool4 equal", None if outs[0][3] is None else np.array_equal(outs[0][3], outs[1][3]),
              "maxdiff", float(np.abs(outs[0][0] - outs[1][0]).max()), flush=True)
