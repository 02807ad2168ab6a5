"""Phase cycles of conv12_ws_kernel (PSEG_WS_TRACE=1): per-wave s_memtime sums written by the kernel to
gpurun_out/ws_trace.bin -- consumers: k-loop / epilogue / barrier wait, producers: fill / barrier wait."""
import os, sys
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")   # PSEG_* of the environment -> plan switches of the engines created here
os.environ["PSEG_WS_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "page-segmentation_amd")]
import numpy as np
import torch
torch.cuda.is_available()
import pseg_amd
from pseg_amd import synth
H, W = 2048, 1536
eng = pseg_amd.Engine("fcn_skip", 3, mode=pseg_amd.MODE_BF16)
eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
img = torch.from_numpy(synth.synth_page(0, H, W, 3)[0]).cuda()
lab = torch.empty((H, W), dtype=torch.uint8, device="cuda")
os.makedirs("gpurun_out", exist_ok=True)
for _ in range(3):
    eng.predict_device(img.data_ptr(), H, W, d_labels_u8=lab.data_ptr())
torch.cuda.synchronize()
t = np.fromfile("gpurun_out/ws_trace.bin", dtype=np.uint64).reshape(-1, 8, 4).astype(np.float64)
n = t[:, :, 3]
cons, prod = t[:, :4], t[:, 4:]
print("tiles per workgroup: min %d max %d" % (n.min(), n.max()))
print("consumer per tile: k-loop %.0f  epilogue %.0f  barrier wait %.0f cycles (sum %.0f)" % (
    (cons[..., 0] / cons[..., 3]).mean(), (cons[..., 1] / cons[..., 3]).mean(), (cons[..., 2] / cons[..., 3]).mean(),
    (cons[..., :3].sum(-1) / cons[..., 3]).mean()))
print("producer per tile: fill %.0f  barrier wait %.0f cycles (sum %.0f)" % (
    (prod[..., 0] / prod[..., 3]).mean(), (prod[..., 2] / prod[..., 3]).mean(), ((prod[..., 0] + prod[..., 2]) / prod[..., 3]).mean()))
print("per-wave k-loop spread:", np.percentile(cons[..., 0] / cons[..., 3], [5, 50, 95]).round(0))
print("per-wave fill spread:", np.percentile(prod[..., 0] / prod[..., 3], [5, 50, 95]).round(0))
