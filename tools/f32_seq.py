"""Per-kernel durations of one float32 predict under rocprofv3 (run as: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/f32_seq.py run), then print: python3 tools/f32_seq.py show DIR"""
import sys, os, glob, csv
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")   # PSEG_* of the environment -> plan switches of the engines created here
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "run":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "page-segmentation_amd")]
    import torch; torch.cuda.is_available()
    import pseg_amd as gpu
    from pseg_amd import synth
    e = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_F32_EXACT)
    e.set_weights(synth.glorot_weights(e.weight_specs(), seed=1))
    img = synth.synth_page(0, 2048, 1536, 3)[0]
    for _ in range(4): e.predict(img, want_logits=False, want_probs=False)
else:
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[-1]
    ks = [r for r in csv.DictReader(open(f)) if "pseg::" in r["Kernel_Name"]]
    print(" ".join("%.0f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in ks[-18:]))
