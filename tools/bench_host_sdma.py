"""Host path (pinned pages in, pinned label maps out) under runtime copy-engine settings: does the device -> host copy leave the
compute units alone when it goes through SDMA instead of a blit kernel?  python tools/bench_host_sdma.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path[:0] = [%r, os.path.join(%r, "page-segmentation_amd")]
import numpy as np, torch
torch.cuda.is_available()
import pseg_amd
from pseg_amd import synth
H, W, C, N = 2048, 1536, 3, 32
eng = pseg_amd.Engine("fcn_skip", C, mode=pseg_amd.MODE_BF16)
eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
base = [synth.synth_page(100 + i, H, W, C)[0] for i in range(4)]
pages = [pseg_amd.pinned_copy(base[i %% 4]) for i in range(N)]
outs = [pseg_amd.pinned_empty((H, W), np.uint8) for _ in range(N)]
eng.predict_batch(pages, dtype=np.uint8, out=outs)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); eng.predict_batch(pages, dtype=np.uint8, out=outs); ts.append((time.perf_counter() - t0) / N)
print("%%.4f ms/page  %%.0f Mpx/s" %% (np.median(ts) * 1e3, H * W / np.median(ts) / 1e6))
''' % (ROOT, ROOT)
for env in ({}, {"HSA_ENABLE_SDMA": "1"}, {"HSA_ENABLE_SDMA": "0"}, {"GPU_FORCE_BLIT_COPY_SIZE": "0"}, {"HIP_FORCE_DEV_KERNARG": "1"},
            {"ROC_ACTIVE_WAIT_TIMEOUT": "0"}, {"GPU_MAX_HW_QUEUES": "8"}):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=300)
    print(env or "default", (r.stdout.strip().splitlines() or ["?"])[-1], ("ERR " + r.stderr.strip().splitlines()[-1][:120]) if r.returncode else "", flush=True)
