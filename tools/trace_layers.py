"""Per-workgroup phase breakdown (s_memtime stamps) of the bf16 conv kernels, one layer at a time.
Needs the diagnostic build:  make -C page-segmentation_amd/csrc libpseg_diag.so
    PSEG_LIB=page-segmentation_amd/csrc/libpseg_diag.so python tools/trace_layers.py [arch] [layers...]"""
import os, sys, subprocess
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")   # PSEG_* of the environment -> plan switches of the engines created here
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "page-segmentation_amd")]
import numpy as np
import torch
torch.cuda.is_available()
from pseg_amd import engine as E, synth

arch = sys.argv[1] if len(sys.argv) > 1 else "fcn_skip"
layers = sys.argv[2:] or ["conv2d_1", "conv2d_2", "conv2d_3", "conv2d_4", "conv2d_5", "conv2d_transpose_2", "conv2d_transpose_4"]
H, W = 2048, 1536
img = torch.from_numpy(synth.synth_page(1000, H, W, 3)[0]).cuda()
lab = torch.empty((H, W), dtype=torch.uint8, device="cuda")
os.makedirs("gpurun_out", exist_ok=True)
for ly in layers:
    os.environ["PSEG_TRACE"] = ly          # the PSEG_* knobs are read once per engine creation
    eng = E.Engine(arch, 3, device=0, mode=E.MODE_BF16)
    eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
    for _ in range(3):                     # (the last traced call's file is the one read below)
        eng.predict_device(img.data_ptr(), H, W, d_labels_u8=lab.data_ptr())
    torch.cuda.synchronize()
    eng.close()
    os.environ.pop("PSEG_TRACE")
    fn = "gpurun_out/trace_%s.bin" % ly
    if not os.path.exists(fn):
        print(ly, ": no trace written"); continue
    a = np.fromfile(fn, dtype=np.uint64).reshape(-1, 12)
    t = a[:, :7].astype(np.int64)
    t_ok = t[:, 6] > 0
    t = t[t_ok]
    d = np.diff(t, axis=1)
    names = ["ring issue", "tab fetch", "tile stage", "wait+barrier", "k-loop", "epilogue"]
    tot = np.median(t[:, 6] - t[:, 0])
    span = t[:, 6].max() - t[:, 0].min()
    print("%-20s WGs %5d  total/WG %6d  kernel span %8d ticks | " % (ly, len(t), tot, span) +
          "  ".join("%s %d" % (n, np.median(d[:, i])) for i, n in enumerate(names)) +
          "  | in k-loop: group waits %d  weight-DMA issue %d" % (np.median(a[t_ok, 8].astype(np.int64)), np.median(a[t_ok, 9].astype(np.int64))))
