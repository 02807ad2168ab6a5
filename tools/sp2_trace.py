"""Phase stamps of conv_sp2_kernel (diagnostic build; PSEG_SP_TRACE=<layer> + PSEG_SP2_TRACE=1), one layer at a time, 2048x1536 fcn_skip page:
    PSEG_LIB=page-segmentation_amd/csrc/libpseg_diag.so python tools/sp2_trace.py [layers...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "page-segmentation_amd")]
import numpy as np
import torch
torch.cuda.is_available()
from pseg_amd import engine as E, synth
layers = [x for x in sys.argv[1:]] or ["conv2d_4", "conv2d_5", "conv2d_transpose_2"]
H, W = 2048, 1536
img = torch.from_numpy(synth.synth_page(1000, H, W, 3)[0]).cuda()
lab = torch.empty((H, W), dtype=torch.uint8, device="cuda")
os.makedirs("gpurun_out", exist_ok=True)
os.environ["PSEG_SP2_TRACE"] = "1"
for ly in layers:
    os.environ["PSEG_SP_TRACE"] = ly
    eng = E.Engine("fcn_skip", 3, device=0, mode=E.MODE_BF16)
    eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
    for _ in range(3):
        eng.predict_device(img.data_ptr(), H, W, d_labels_u8=lab.data_ptr())
    torch.cuda.synchronize()
    eng.close()
    os.environ.pop("PSEG_SP_TRACE")
    a = np.fromfile("gpurun_out/sp2_trace_%s.bin" % ly, dtype=np.uint64).reshape(-1, 16).astype(np.int64)
    t0 = a[:, 0]
    med = lambda x: int(np.median(x))
    r = lambda i: a[:, i] - t0
    print("%-20s WGs %3d | team0: ready %d kloop-end %d epi-end %d | tile1: ready %d kloop-end %d epi-end %d | end %d slow-wait %d (for weights %d) || team1 tile0: ready %d kloop-end %d slow-wait %d || weight loader polling for ring space %d | tile loader end %d weight loader end %d | launch span %d" % (
        ly, len(a), med(r(1)), med(r(2)), med(r(3)), med(r(4)), med(r(5)), med(r(6)), med(r(7)), med(a[:, 11]), med(a[:, 15]),
        med(r(8)), med(r(9)), med(a[:, 12]), med(a[:, 10]), med(r(13)), med(r(14)), int(a[:, 7].max() - t0.min())), flush=True)
