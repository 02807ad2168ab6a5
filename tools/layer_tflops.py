"""Per-layer time and algorithmic TFLOP/s of one bf16 predict (bench.py timing slots)."""
import json, subprocess, sys
arch = sys.argv[1] if len(sys.argv) > 1 else "unet"
out = subprocess.run([sys.executable, "bench.py", "--arch", arch, "--no-cpu-baseline", "--steps", "5"], capture_output=True, text=True).stdout.strip().split("\n")[-1]
d = json.loads(out)
print(d["ms_per_step"], "ms/page  whole-net frac", d["roofline"]["whole_net_frac"])
sys.path[:0] = [".", "page-segmentation_amd"]
import torch; torch.cuda.is_available()
from pseg_amd import engine as E
eng = E.Engine(arch, 3, mode=E.MODE_BF16)
import numpy as np
from pseg_amd import synth
eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=1))
img = synth.synth_page(1000, 2048, 1536, 3)[0]
for _ in range(3): eng.predict(img, want_logits=False, want_probs=False)   # first launches load code objects
eng.timing_enable(True); eng.timing_reset()
for _ in range(5): eng.predict(img, want_logits=False, want_probs=False)
for name, ms, n, flops in eng.timing():
    if n: print("%-22s %8.1f us  %7.1f GFLOP  %7.1f TFLOP/s" % (name, 1e3 * ms / n, flops / 1e9, flops / (ms / n * 1e-3) / 1e12))
