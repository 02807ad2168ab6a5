"""conv_sp_kernel against the conv_mfma_kernel instances it replaces (PSEG_NO_SP=1): stored activations layer by layer, labels,
per-kernel times.  GPU box:  python tools/dbg_sp.py [H W]..."""
import os, sys
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")   # PSEG_* of the environment -> plan switches of the engines created here
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "page-segmentation_amd")]
import numpy as np
import torch
torch.cuda.is_available()   # (torch initialises the device first)
from pseg_amd import engine as E, synth

def make(arch, C, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    eng = E.Engine(arch, C, device=0, mode=E.MODE_BF16)
    for k, v in old.items():
        if v is None: os.environ.pop(k)
        else: os.environ[k] = v
    eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
    return eng

args = [x for x in sys.argv[1:] if not x.startswith("--")]
sizes = [(int(args[i]), int(args[i + 1])) for i in range(0, len(args) - 1, 2)] or [(96, 80), (256, 320), (130, 67), (2048, 1536)]
bad = 0
for arch, C in (() if "--time-only" in sys.argv else (("fcn_skip", 3), ("fcn", 3), ("fcn_skip", 6))):
    a = make(arch, C, {"PSEG_SP_CHECK": "1"})
    b = make(arch, C, {"PSEG_NO_SP": "1"})
    for H, W in sizes:
        img = synth.synth_page(1000, H, W, C)[0]
        la, _, pa = a.predict(img, want_probs=False)
        lb, _, pb = b.predict(img, want_probs=False)
        dl = float(np.abs(la - lb).max())
        line = "%-9s C=%d %4dx%-4d  max|dlogit| %.3g (|logit| max %.3g)  labels differ %d" % (arch, C, H, W, dl, float(np.abs(lb).max()), int((pa != pb).sum()))
        for ly in ("conv2d_4", "conv2d_5", "conv2d_6", "conv2d_transpose_1", "conv2d_transpose_2"):
            try:
                xa, xb = a.activation(ly), b.activation(ly)
            except Exception as ex:
                line += "  %s: n/a" % ly
                continue
            d = float(np.abs(xa.astype(np.float32) - xb.astype(np.float32)).max())
            line += "  %s: %s%.3g/%.3g" % (ly, "EQ " if np.array_equal(xa, xb) else "", d, float(np.abs(xb).max()))
            if d > 0.03 * max(1.0, float(np.abs(xb).max())): bad += 1
        print(line, flush=True)
        if dl > 0.03 * max(1.0, float(np.abs(lb).max())): bad += 1
    a.close(); b.close()
print("BAD" if bad else "OK", bad)
if True:
  for H, W in ((2048, 1536), (4096, 3072), (1024, 768)):
    img = torch.from_numpy(synth.synth_page(1000, H, W, 3)[0]).cuda()
    lab = torch.empty((H, W), dtype=torch.uint8, device="cuda")
    for env in ({}, {"PSEG_NO_SP": "1"}):
        eng = make("fcn_skip", 3, env)
        for _ in range(5): eng.predict_device(img.data_ptr(), H, W, d_labels_u8=lab.data_ptr())
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        s = torch.cuda.ExternalStream(eng.stream())
        with torch.cuda.stream(s):
            t0.record()
            for _ in range(20): eng.predict_device(img.data_ptr(), H, W, d_labels_u8=lab.data_ptr())
            t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 20
        eng.timing_enable(True); eng.timing_reset()
        for _ in range(10): eng.predict_device(img.data_ptr(), H, W, d_labels_u8=lab.data_ptr())
        torch.cuda.synchronize()
        tm = eng.timing()
        print("%dx%d" % (H, W), env or "default", "%.4f ms/page" % ms, " ".join("%s=%.1f" % (k[7:] or "c", t / max(n, 1) * 1e3) for k, t, n, _ in tm), flush=True)
        eng.close()
sys.exit(1 if bad else 0)
