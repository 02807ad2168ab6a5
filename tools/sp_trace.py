"""Phase stamps of conv_sp_kernel (diagnostic build, PSEG_SP_TRACE=<layer>), one layer at a time, 2048x1536 fcn_skip page:
    PSEG_LIB=page-segmentation_amd/csrc/libpseg_diag.so python tools/sp_trace.py [layers...]"""
import os, sys
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")   # PSEG_* of the environment -> plan switches of the engines created here
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "page-segmentation_amd")]
import numpy as np
import torch
torch.cuda.is_available()
from pseg_amd import engine as E, synth
layers = [x for x in sys.argv[1:]] or ["conv2d_4", "conv2d_5", "conv2d_6", "conv2d_transpose", "conv2d_transpose_2"]
H, W = 2048, 1536
img = torch.from_numpy(synth.synth_page(1000, H, W, 3)[0]).cuda()
lab = torch.empty((H, W), dtype=torch.uint8, device="cuda")
os.makedirs("gpurun_out", exist_ok=True)
for ly in layers:
    os.environ["PSEG_SP_TRACE"] = ly
    eng = E.Engine("fcn_skip", 3, device=0, mode=E.MODE_BF16)
    eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
    for _ in range(3):
        eng.predict_device(img.data_ptr(), H, W, d_labels_u8=lab.data_ptr())
    torch.cuda.synchronize()
    eng.close()
    os.environ.pop("PSEG_SP_TRACE")
    a = np.fromfile("gpurun_out/sp_trace_%s.bin" % ly, dtype=np.uint64).reshape(-1, 16).astype(np.int64)
    t0 = a[:, 0]
    med = lambda x: int(np.median(x))
    r = lambda i: a[:, i] - t0
    print("%-20s WGs %3d | barrier %d  tile0: ready %d kloop-end %d epi-end %d | tile1: ready %d kloop-end %d epi-end %d | end %d  slow-wait cycles %d (%d waits)  | weight loader end %d (polling %d)  tile loader end %d | span %d" % (
        ly, len(a), med(r(1)), med(r(2)), med(r(3)), med(r(4)), med(r(5)), med(r(6)), med(r(7)), med(r(10)), med(a[:, 8]), med(a[:, 9]),
        med(r(11)), med(a[:, 12]), med(r(13)), int(a[:, 10].max() - t0.min())), flush=True)
    hw = np.fromfile("gpurun_out/sp_trace_%s.bin" % ly, dtype=np.uint16).reshape(-1, 64)[:, 56:64]
    simd = (hw >> 4) & 3
    from collections import Counter
    print("   SIMD of waves 0..7 (most common placements):", Counter(tuple(int(x) for x in r_) for r_ in simd).most_common(4), flush=True)
