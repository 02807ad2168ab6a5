"""Debug aid: per-tensor gradient error of the unet train step against torch autograd (float32 and float64 referees)."""
import os, sys
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")   # PSEG_* of the environment -> plan switches of the engines created here
ROOT = os.environ.get("DBG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "page-segmentation_amd")]
import numpy as np
import oracle
from oracle.train_ref import graph_loss_and_grads
from pseg_amd import synth
import pseg_amd as gpu

arch, C, shape = "unet", 3, (32, 64)
img, _, mask = synth.synth_page(2, 96, 96, C)
img = np.ascontiguousarray(img[:shape[0], :shape[1]]); mask = np.ascontiguousarray(mask[:shape[0], :shape[1]])
Wt = oracle.init_weights(arch, C, seed=11, gain=1.2, bias_scale=0.05)
eng = gpu.Engine(arch, C, mode=gpu.MODE_F32_EXACT)
eng.set_weights(Wt)
eng.train_init(clipnorm=1.0)
eng.train_set_dropout_seed(77)
for step in range(2):
    lo, go, _ = graph_loss_and_grads(arch, Wt, img, mask, drop=(77, step), float64=True)
    loss = eng.train_forward_backward(img, mask)[0]
    g = eng.gradients()
    rows = []
    for k in go:
        sc = np.abs(go[k]).max() + 1e-12
        d = np.abs(g[k] - go[k])
        rows.append((float(d.max() / sc), k, int((d > 1e-4 * sc).sum()), d.size))
    print("step", step, "loss", loss, lo, "knobs", {k: v for k, v in os.environ.items() if k.startswith("PSEG_")})
    for r in sorted(rows, reverse=True)[:8]:
        print("   %.3e %-22s elems>1e-4: %d / %d" % r)

# activations after the train step vs the oracle's forward (layers ahead of the Dropout sites are unaffected by it)
z, acts = oracle.forward(arch, Wt, img, "f32", return_acts=True)
for name in ("conv2d", "conv2d_1", "conv2d_2", "conv2d_3", "conv2d_4", "conv2d_5", "conv2d_6"):
    a = eng.activation(name)
    o = acts[name]
    d = np.abs(a - o)
    print("act %-10s shape %s equal %s  max|d| %.3e  n_diff %d" % (name, a.shape, np.array_equal(a, o), d.max(), int((d > 0).sum())))
eng2 = gpu.Engine(arch, C, mode=gpu.MODE_F32_EXACT)
eng2.set_weights(Wt)
eng2.predict(img)
for name in ("conv2d_3", "conv2d_4", "conv2d_5"):
    print("predict act %-10s equal oracle: %s" % (name, np.array_equal(eng2.activation(name), acts[name])))
