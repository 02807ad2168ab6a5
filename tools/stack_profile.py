"""Per-layer MFMA utilisation from a rocprofv3 kernel trace (no in-process timers: event pairs between layers
inflate the numbers).  Two steps inside one gpurun command:
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/stack -- python3 tools/stack_profile.py run unet
  python3 tools/stack_profile.py merge unet gpurun_out/stack profiles/<name>.json
`run` predicts PAGES pages of 2048x1536 and writes the launch-order layer table (name, kernel size, algorithmic
flops) to gpurun_out/stack_ops_<arch>.json; `merge` pairs it with the trace and reports every layer plus the
k3-conv-stack aggregate (north star: >= 40 % of the dense bf16 MFMA peak on the 3x3 stack of unet)."""
import csv, glob, json, os, sys
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")   # PSEG_* of the environment -> plan switches of the engines created here
PAGES, PEAK = 12, 2.5e15
mode, arch = sys.argv[1], sys.argv[2]
if mode == "run":
    sys.path[:0] = [".", "page-segmentation_amd"]
    import numpy as np
    import torch; torch.cuda.is_available()
    from pseg_amd import engine as E, synth
    eng = E.Engine(arch, 3, mode=E.MODE_BF16)
    eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=1))
    ksize = {n.split("/")[0]: s[0] for n, s in eng.weight_specs() if n.endswith("kernel")}
    img = synth.synth_page(1000, 2048, 1536, 3)[0]
    eng.timing_enable(True); eng.timing_reset()
    eng.predict(img, want_logits=False, want_probs=False)          # one timed page: the slot table (names, flops)
    slots = [(name, ksize.get(name, 0), flops) for name, ms, n, flops in eng.timing() if n]
    eng.timing_enable(False)
    out = np.empty((2048, 1536), np.uint8)
    for _ in range(PAGES): eng.predict(img, want_logits=False, want_probs=False)
    json.dump({"slots": slots, "pages": PAGES}, open("gpurun_out/stack_ops_%s.json" % arch, "w"))
else:
    tdir, dst = sys.argv[3], sys.argv[4]
    ops = json.load(open("gpurun_out/stack_ops_%s.json" % arch))
    slots, pages = ops["slots"], ops["pages"]
    rows = []
    for f in glob.glob(os.path.join(tdir, "**", "*kernel_trace.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "pseg::" in r["Kernel_Name"] or r["Kernel_Name"].startswith("Cijk_")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a library GEMM (hipBLASLt "Cijk_...") belongs to the split up-conv layer whose gather-sum pass follows it
    merged = []
    carry = 0
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if r["Kernel_Name"].startswith("Cijk_"):
            carry += d
            continue
        merged.append({"Kernel_Name": ("GEMM + " if carry else "") + r["Kernel_Name"], "Start_Timestamp": 0, "End_Timestamp": d + carry})
        carry = 0
    rows = merged
    n = len(slots)
    rows = rows[-n * (pages - 2):]                                 # the last pages - 2 untimed pages
    assert len(rows) == n * (pages - 2), (len(rows), n)
    dur = [0.0] * n
    for i, r in enumerate(rows):
        dur[i % n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9 / (pages - 2)
    layers = [{"layer": s[0], "k": s[1], "gflop": s[2] / 1e9, "us": d * 1e6, "tflops": s[2] / d / 1e12, "frac": s[2] / d / PEAK,
               "kernel": rows[i]["Kernel_Name"][:70]} for i, (s, d) in enumerate(zip(slots, dur))]
    k3 = [l for l in layers if l["k"] == 3 and l["gflop"] > 10]   # the first layer (Cin = 1, 3.6 GFLOP) is a write stream, not MFMA work
    agg = lambda ls: {"gflop": sum(l["gflop"] for l in ls), "us": sum(l["us"] for l in ls),
                      "frac": sum(l["gflop"] for l in ls) * 1e9 / (sum(l["us"] for l in ls) * 1e-6) / PEAK}
    res = {"arch": arch, "page": "2048x1536", "peak_tflops": PEAK / 1e12, "source": "rocprofv3 --kernel-trace, mean of %d pages" % (pages - 2),
           "whole_net": agg(layers), "conv3x3_stack": agg(k3), "layers": layers}
    json.dump(res, open(dst, "w"), indent=1)
    for l in layers: print("%-12s k%d %8.1f us %7.1f GFLOP %7.1f TFLOP/s %5.1f %%" % (l["layer"], l["k"], l["us"], l["gflop"], l["tflops"], 100 * l["frac"]))
    print("whole net (kernel time) %.1f us  %.1f %%   3x3 conv stack %.1f us  %.1f %%" % (res["whole_net"]["us"], 100 * res["whole_net"]["frac"], res["conv3x3_stack"]["us"], 100 * res["conv3x3_stack"]["frac"]))
