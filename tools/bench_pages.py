"""Page batches (pseg_predict_pages_device / pseg_predict_batch units) against page-by-page launches: labels equal, ms per page.
    python tools/bench_pages.py [pages] [H W]"""
import os, sys, time
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")   # PSEG_* of the environment -> plan switches of the engines created here
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "page-segmentation_amd")]
import numpy as np
import torch
torch.cuda.is_available()
import pseg_amd
from pseg_amd import engine as E, synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2048, 1536)
C = 3
eng = E.Engine("fcn_skip", C, device=0, mode=E.MODE_BF16)
eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
host = np.stack([synth.synth_page(1000 + i, H, W, C)[0] for i in range(P)])
d = torch.from_numpy(host).cuda()
one = torch.empty((P, H, W), dtype=torch.uint8, device="cuda")
st = eng.stream()
s = torch.cuda.ExternalStream(st)
def per_page():
    for i in range(P):
        eng.predict_device(d[i].data_ptr(), H, W, d_labels_u8=one[i].data_ptr(), stream=st)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            t0.record(); fn(); t1.record()
        torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) / P)
    return float(np.median(ts))
t_one = timed(per_page)
ref = one.cpu().numpy().copy()
print("page by page: %.4f ms/page" % t_one, flush=True)
for cap in (2, 4, 8, 16, 32):
    if cap > P: break
    os.environ["PSEG_BATCH_PAGES"] = str(cap)
    e2 = E.Engine("fcn_skip", C, device=0, mode=E.MODE_BF16)
    os.environ.pop("PSEG_BATCH_PAGES")
    e2.set_weights(synth.glorot_weights(e2.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
    out = torch.zeros((P, H, W), dtype=torch.uint8, device="cuda")
    st2 = e2.stream(); s2 = torch.cuda.ExternalStream(st2)
    def batched():
        e2.predict_pages_device(d.data_ptr(), P, H, W, d_labels_u8=out.data_ptr(), stream=st2)
    batched(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s2):
            t0.record(); batched(); t1.record()
        torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) / P)
    eq = bool(np.array_equal(out.cpu().numpy(), ref))
    e2.timing_enable(True); e2.timing_reset(); batched(); torch.cuda.synchronize()
    tm = " ".join("%s=%.1f" % (k[7:] or "c", t / P * 1e3) for k, t, n, _ in e2.timing() if n)
    print("units of %2d pages: %.4f ms/page (%.1f %% of page by page)  labels equal: %s | us per page: %s" % (cap, float(np.median(ts)), 100 * float(np.median(ts)) / t_one, eq, tm), flush=True)
    e2.close()
# host path: pinned pages in, pinned maps out
pages = [pseg_amd.pinned_copy(host[i]) for i in range(min(P, 16))]
outs = [pseg_amd.pinned_empty((H, W), np.uint8) for _ in pages]
for cap in (1, 8):
    os.environ["PSEG_BATCH_PAGES"] = str(cap)
    e3 = E.Engine("fcn_skip", C, device=0, mode=E.MODE_BF16)
    os.environ.pop("PSEG_BATCH_PAGES")
    e3.set_weights(synth.glorot_weights(e3.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
    e3.predict_batch(pages, dtype=np.uint8, out=outs)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); e3.predict_batch(pages, dtype=np.uint8, out=outs); ts.append((time.perf_counter() - t0) / len(pages))
    eq = all(np.array_equal(np.asarray(o), ref[i]) for i, o in enumerate(outs))
    print("pseg_predict_batch, pinned in/out, units of %d: %.4f ms/page  labels equal: %s" % (cap, float(np.median(ts)) * 1e3, eq), flush=True)
    e3.close()
