"""Random shapes / class counts through conv_sp2_kernel (PSEG_SP2=24 / 32) against conv_mfma_kernel (PSEG_NO_SP): logits and labels bit for bit.
    python tools/fuzz_sp2.py   (24 trials, 0 mismatches in round 5)"""
import os, sys
os.environ["PSEG_PLAN_FROM_ENV"] = "1"
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "page-segmentation_amd")]
import numpy as np
import pseg_amd as gpu
from pseg_amd import synth
rng = np.random.default_rng(2025)
bad = 0
for trial in range(24):
    arch = ["fcn_skip", "fcn"][trial % 2]
    C = int(rng.integers(2, 9))
    H, W = int(rng.integers(1, 700)), int(rng.integers(1, 900))
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    res = []
    for env in ({"PSEG_SP2": "24", "PSEG_SP_CHECK": "1"}, {"PSEG_SP2": "32", "PSEG_SP_CHECK": "1"}, {"PSEG_NO_SP": "1"}):
        for k in ("PSEG_SP2", "PSEG_SP_CHECK", "PSEG_NO_SP"): os.environ.pop(k, None)
        os.environ.update(env)
        e = gpu.Engine(arch, C, mode=gpu.MODE_BF16)
        e.set_weights(synth.glorot_weights(e.weight_specs(), seed=trial, gain=1.5, bias_scale=0.05))
        z, _, l = e.predict(img, want_probs=False)
        res.append((z, l))
        e.close()
    ok = all(np.array_equal(res[0][0], r[0]) and np.array_equal(res[0][1], r[1]) for r in res[1:])
    bad += not ok
    print(trial, arch, C, (H, W), "ok" if ok else "MISMATCH", flush=True)
print("mismatches:", bad)
