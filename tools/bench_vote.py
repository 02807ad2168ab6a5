"""Vote timing on configs[4]'s page (4096x3072, 6 classes, uint8 labels): the synthetic page as BASELINE.md prescribes it
(two speckled image rectangles: one percolating component each + thousands of specks) and the same page with the
rectangles blanked (text only).  Default: the vote's tile pass (vote_tile_kernel + border unions + root merge + run list);
PSEG_CCL_GLOBAL=1: the page-global union-find over pixel indices with the 32 x 32 counting tiles.  PSEG_LIB=<other build>
times another library on the same box (boxes differ by up to 10 %)."""
import ctypes, json, os, sys, time
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")   # PSEG_* of the environment -> plan switches of the engines created here
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "page-segmentation_amd")]
import numpy as np
import torch
torch.cuda.is_available()
from pseg_amd import engine as E, synth
H, W, C = 4096, 3072, 6
_, binary, mask = synth.synth_page(1000, H, W, C)
text_only = binary.copy(); text_only[mask == 2] = 0
dev = torch.device("cuda:0")
L = E.lib(); vp = ctypes.c_void_p
st = torch.cuda.current_stream().cuda_stream
out = {"knobs": {k: v for k, v in os.environ.items() if k.startswith("PSEG_")}}
for name, b in (("page", binary), ("text_only", text_only)):
    d_b = torch.from_numpy(b).to(dev)
    src = torch.from_numpy(mask.astype(np.uint8)).to(dev)
    d_p = src.clone()
    def run():
        E._check(L.pseg_cc_vote_device_u8(0, vp(d_p.data_ptr()), vp(d_b.data_ptr()), H, W, C, vp(st)))
    for _ in range(3): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize()
    out[name] = {"ms": round((time.perf_counter() - t0) / 20 * 1e3, 4), "ink_frac": round(float(b.mean()), 4)}
print(json.dumps(out))
