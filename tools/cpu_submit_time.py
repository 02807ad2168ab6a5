"""How long the host takes to SUBMIT a page (all launches of pseg_predict_pages_device, no synchronisation) against how long the GPU
takes to run it: tools/cpu_submit_time.py [pages]"""
import os, sys, time
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "page-segmentation_amd")]
import numpy as np
import torch
torch.cuda.is_available()
from pseg_amd import engine as E, synth
H, W = 2048, 1536
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = E.Engine("fcn_skip", 3, mode=E.MODE_BF16)
eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
img = torch.from_numpy(np.stack([synth.synth_page(1000 + i, H, W, 3)[0] for i in range(P)])).cuda()
lab = torch.empty((P, H, W), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(5): eng.predict_pages_device(img.data_ptr(), P, H, W, d_labels_u8=lab.data_ptr(), stream=st)
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for _ in range(N): eng.predict_pages_device(img.data_ptr(), P, H, W, d_labels_u8=lab.data_ptr(), stream=st)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("pages/call %d: host submit %.1f us per page, GPU %.1f us per page" % (P, (t1 - t0) / N / P * 1e6, (t2 - t0) / N / P * 1e6))
