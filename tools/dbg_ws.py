import os, sys, subprocess
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")   # PSEG_* of the environment -> plan switches of the engines created here
sys.path[:0] = ['.', 'page-segmentation_amd']
import numpy as np
if len(sys.argv) > 1:
    import torch; torch.cuda.is_available()
    import pseg_amd, oracle
    rng = np.random.default_rng(77)
    Wt = oracle.init_weights("fcn_skip", 3, seed=4, gain=1.5, bias_scale=0.05)
    for shape in [(16, 32), (96, 80)]:
        im = rng.integers(0, 256, size=shape, dtype=np.uint8)
        e = pseg_amd.Engine("fcn_skip", 3, mode=pseg_amd.MODE_BF16); e.set_weights(Wt)
        z, _, l = e.predict(im, want_probs=False)
        zo = oracle.forward("fcn_skip", Wt, im, "bf16")
        print(sys.argv[1], shape, "max |z - oracle_bf16| = %.4g  (max |z| %.3g)" % (np.abs(z - zo).max(), np.abs(zo).max()))
        e.close()
else:
    subprocess.check_call([sys.executable, __file__, "ws"])
    subprocess.check_call([sys.executable, __file__, "nows"], env=dict(os.environ, PSEG_NO_WS="1"))
    subprocess.check_call([sys.executable, __file__, "nofuse"], env=dict(os.environ, PSEG_NO_CONV1_FUSION="1"))
