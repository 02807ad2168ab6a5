"""BASELINE.json configs[4] ("6-class predict at 4096x3072 with connected-component + overlay/inverted
post-process on GPU") with every buffer resident in HBM: per-stage time and, for the HBM-bound
post-process kernels, algorithmic GB/s against the 8 TB/s peak (SURVEY.md 8d byte counts).
Also times prepare_images (host buffers) on an A4 scan.  Prints one JSON line.
Default: the uint8 label map end to end (pseg_predict_device labels_u8 -> pseg_cc_vote_device_u8 -> pseg_masks_device_u8), the
entries bench.py's extra.config5 leg times and the Predictor chain uses; --int64: the reference's int64 maps."""
import ctypes, json, os, sys, time
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")   # PSEG_* of the environment -> plan switches of the engines created here
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "page-segmentation_amd")]
import numpy as np
import torch
torch.cuda.is_available()
from pseg_amd import engine as E, synth

H, W, C = 4096, 3072, 6
I64 = "--int64" in sys.argv
dev = torch.device("cuda:0")
eng = E.Engine("fcn_skip", C, mode=E.MODE_BF16)
eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
img, binary, _ = synth.synth_page(1000, H, W, C)
d_img = torch.from_numpy(img).to(dev)
d_bin = torch.from_numpy(binary).to(dev)
d_lab = torch.empty((H, W), dtype=torch.int64 if I64 else torch.uint8, device=dev)
lut = torch.from_numpy(np.array([[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255]], np.uint8)).to(dev)
outs = [torch.empty((H, W, 3), dtype=torch.uint8, device=dev) for _ in range(4)]
L = E.lib()
st = torch.cuda.current_stream().cuda_stream
vp = ctypes.c_void_p


def t(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def predict():
    if I64: eng.predict_device(d_img.data_ptr(), H, W, d_labels=d_lab.data_ptr(), stream=st)
    else: eng.predict_device(d_img.data_ptr(), H, W, d_labels_u8=d_lab.data_ptr(), stream=st)


def vote():
    E._check((L.pseg_cc_vote_device if I64 else L.pseg_cc_vote_device_u8)(0, vp(d_lab.data_ptr()), vp(d_bin.data_ptr()), H, W, C, vp(st)))


def masks():
    E._check((L.pseg_masks_device if I64 else L.pseg_masks_device_u8)(0, vp(d_lab.data_ptr()), vp(d_bin.data_ptr()), vp(lut.data_ptr()), C, H, W,
                                 vp(outs[0].data_ptr()), vp(outs[1].data_ptr()), vp(outs[2].data_ptr()), vp(outs[3].data_ptr()), vp(st)))


px = H * W
tp, tv, tm = t(predict), t(vote), t(masks)
res = {"workload": "configs[4]: 4096x3072, 6 classes, fcn_skip bf16 + cc_majority vote + 4 masks, HBM-resident",
       "predict_ms": round(tp * 1e3, 3), "predict_Mpx_s": round(px / tp / 1e6, 1),
       "predict_TFLOPs": round(eng.flops_per_pixel() * px / tp / 1e12, 1),
       "label_map": "int64" if I64 else "uint8",
       "cc_vote_ms": round(tv * 1e3, 3), "cc_vote_alg_GBs": round(px * ((8 + 8 + 1 + 8) if I64 else 4) / tv / 1e9, 1),
       "cc_vote_alg_bytes_per_px": "2 reads + 1 write of the label map + 1 read of uint8 binary = %d" % (25 if I64 else 4),
       "masks_ms": round(tm * 1e3, 3), "masks_alg_GBs": round(px * ((8 + 1 + 12) if I64 else 14) / tm / 1e9, 1),
       "masks_alg_bytes_per_px": "label map + binary in, 4 x 3 out = %d" % (21 if I64 else 14),
       "pipeline_ms": round((tp + tv + tm) * 1e3, 3), "pipeline_Mpx_s": round(px / (tp + tv + tm) / 1e6, 1), "hbm_peak_GBs": 8000}
# loader side (host buffers, includes PCIe): A4 scan at 300 dpi, line height 25 -> 6
scan, sbin = synth.synth_page(7, 3508, 2480, 3)[:2]
scan = 255 - scan
sbin255 = np.where(scan > 127, 255, 0).astype(np.uint8)
for _ in range(2):
    E.prepare_images(scan, sbin255, 6 / 25)
t0 = time.perf_counter()
for _ in range(5):
    E.prepare_images(scan, sbin255, 6 / 25)
tl = (time.perf_counter() - t0) / 5
res["prepare_images_ms_A4_host_buffers"] = round(tl * 1e3, 2)
res["prepare_images_Mpx_s"] = round(scan.size / tl / 1e6, 1)
print(json.dumps(res))
