#!/usr/bin/env python3
"""bench.py -- Mpixels/s classified on synthetic 2048x1536 3-class pages (BASELINE.json).

One "step" = one pass of the predict hot path (x/255 -> pad -> fcn_skip -> crop -> logits ->
argmax) over `--pages` synthetic pages per rank, inputs already resident in HBM, uint8 label
maps left in HBM (the task's measurement contract; the host-buffer / PCIe-inclusive rate of the
drop-in boundary -- SURVEY.md 8d's "uint8 page in pinned host memory -> label map in host memory" -- is
reported next to it at the top level as `value_host_path` and in detail under `extra.host_path`, never as
`value`).  N=1 runs BASELINE.json configs[1] (single 2048x1536 page, 3 classes, bf16 activations).  N>1:
one process per GPU, BASELINE.json configs[2]'s shape -- 32 independent pages per rank per step (`--pages`),
no data-path collective (weak scaling); value = pixels of all ranks / max-over-ranks time.

`python bench.py --gpus N` with WORLD_SIZE unset starts the N rank processes itself (a parent
that never touches the GPU spawns fresh children with RANK / LOCAL_RANK / WORLD_SIZE set and
relays rank 0's line); under torchrun (WORLD_SIZE set) the process is one rank.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, algorithmic FLOPs / HIP-event
duration measured here), `cpu_baseline` (the float32 restatement on torch-CPU / oneDNN at n = all
cores and n = 1; the reference's TensorFlow path cannot run offline) and
`extra` (host path from pinned memory, float32 engine, configs[3] train step, label-exact mode, unet, res_unet,
configs[4] pipeline, the drop-in Predictor chain).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "page-segmentation_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md: dense MFMA peaks
HBM_PEAK_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=("bf16", "f32"), default="bf16")
    ap.add_argument("--arch", default="fcn_skip")
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--height", type=int, default=2048)
    ap.add_argument("--width", type=int, default=1536)
    ap.add_argument("--page-by-page", action="store_true", help="with --pages > 1: one pseg_predict_device call per page instead of pseg_predict_pages_device (A/B)")
    ap.add_argument("--pages", type=int, default=None, help="pages per rank per step (default: 1 at --gpus 1 = configs[1]; 32 at --gpus N > 1 = configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra.* legs (host path, unet, configs[4], label-exact)")
    args = ap.parse_args(argv)
    if args.pages is None:
        args.pages = 1 if args.gpus <= 1 else 32
    return args


# ---------------------------------------------------------------------------------------------------
# parent: start one fresh process per GPU (no GPU call, no torch import in this process)
# ---------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
    # A rank that dies (no such GPU, RCCL refusal, ...) must not leave the others waiting in the rendezvous: poll, and when one
    # has exited with an error stop the rest -- by their own PIDs -- and fail loudly.
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad
            break
        if all(c is not None for c in codes):
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write("bench.py: rank(s) failed %r (rank, exit code); the other ranks were stopped.  --gpus %d needs %d visible GPUs.\n"
                         % (failed, args.gpus, args.gpus))
        return 1
    reader.join(timeout=10)
    line = None
    for ln in (buf[0] if buf else b"").decode("utf-8", "replace").splitlines():
        if ln.startswith("{"):
            line = ln
    if line is None:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    print(line)
    return 0


# ---------------------------------------------------------------------------------------------------
# extra legs (rank 0, N = 1 only; outside the headline timed region)
# ---------------------------------------------------------------------------------------------------
def _sync_time(torch, fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def leg_host_path(np, pseg_amd, eng, synth, H, W, C, n_pages=32, reps=3):
    """SURVEY.md 8d boundary: uint8 pages in PINNED host memory -> label maps in pinned host memory through
    pseg_predict_batch (copies of the neighbouring page units overlap the compute of the current one); configs[2]'s 32 pages
    per rank: eight synthetic pages, each in the list four times (own input and output buffers)."""
    base = [synth.synth_page(100 + i, H, W, C)[0] for i in range(min(8, n_pages))]
    pages = [pseg_amd.pinned_copy(base[i % len(base)]) for i in range(n_pages)]
    out = {}
    for name, dt in (("uint8", np.uint8), ("int64", np.int64)):
        outs = [pseg_amd.pinned_empty((H, W), dt) for _ in range(n_pages)]
        eng.predict_batch(pages, dtype=dt, out=outs)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            eng.predict_batch(pages, dtype=dt, out=outs)
            ts.append((time.perf_counter() - t0) / n_pages)
        ts.sort()
        out[name] = {"ms_per_page": round(ts[len(ts) // 2] * 1e3, 4), "Mpixels_s": round(H * W / ts[len(ts) // 2] / 1e6, 1)}
    # pageable NumPy arrays through the library's pinned staging ring
    pg = [np.array(p) for p in pages]
    outs = [np.empty((H, W), np.uint8) for _ in range(n_pages)]
    eng.predict_batch(pg, dtype=np.uint8, out=outs)
    t0 = time.perf_counter()
    eng.predict_batch(pg, dtype=np.uint8, out=outs)
    tp = (time.perf_counter() - t0) / n_pages
    out["uint8_pageable_via_ring"] = {"ms_per_page": round(tp * 1e3, 4), "Mpixels_s": round(H * W / tp / 1e6, 1)}
    out["what"] = ("pseg_predict_batch, %d pages of %dx%d (configs[2]'s share of one rank), pages and label maps in pinned host memory "
                   "(pseg_host_alloc), units of up to 8 same-shape pages, median of %d passes; PCIe-inclusive, never the headline value" % (n_pages, H, W, reps))
    return out


def leg_pages32(torch, np, eng, synth, H, W, C, dev, steps, warmup, n_pages=32):
    """configs[2]'s share of ONE rank, HBM-resident: 32 pages per step through pseg_predict_pages_device, the same
    timed-region protocol as the headline (W warm-up steps, K steps between two synchronisations).  This is the per-rank
    baseline of the `--gpus N` line (which runs exactly this per rank): scaling efficiency N x this, not N x `value`
    (one page per step: no page units, +7 % per page)."""
    d_pages = torch.from_numpy(np.stack([synth.synth_page(p, H, W, C)[0] for p in range(n_pages)])).to(dev)
    d_labels = torch.empty((n_pages, H, W), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    fn = lambda: eng.predict_pages_device(d_pages.data_ptr(), n_pages, H, W, d_labels_u8=d_labels.data_ptr(), stream=st)
    steps = max(3, min(steps, 10))
    t = _sync_time(torch, fn, steps, warm=max(1, min(warmup, 3)))
    eng.status(st)
    return {"Mpixels_s": round(n_pages * H * W / t / 1e6, 1), "ms_per_page": round(t / n_pages * 1e3, 4), "ms_per_step": round(t * 1e3, 4),
            "pages_per_step": n_pages, "steps": steps,
            "what": "HBM-resident, one pseg_predict_pages_device call per step, pages 0..%d of the synthetic set (rank 0's share of configs[2]); "
                    "the per-rank baseline for the --gpus N line" % (n_pages - 1)}


def leg_arch(torch, pseg_amd, synth, arch, H, W, C, dev, steps=5):
    """ms/page and roofline fractions of another graph (unet = the 3x3 stack north_star's 40 % names) on the same page."""
    eng = pseg_amd.Engine(arch, C, device=dev.index, mode=pseg_amd.MODE_BF16)
    eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
    img = torch.from_numpy(synth.synth_page(0, H, W, C)[0]).to(dev)
    lab = torch.empty((H, W), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    fn = lambda: eng.predict_device(img.data_ptr(), H, W, d_labels_u8=lab.data_ptr(), stream=st)
    t = _sync_time(torch, fn, steps)
    eng.timing_enable(True)
    eng.timing_reset()
    for _ in range(3):
        fn()
    torch.cuda.synchronize(dev)
    slots = [s for s in eng.timing() if s[2] > 0]
    eng.timing_enable(False)
    peak = PEAK_TFLOPS["bf16"]
    res = {"ms_per_page": round(t * 1e3, 4), "Mpixels_s": round(H * W / t / 1e6, 1),
           "whole_net_frac": round(eng.flops_per_pixel() * H * W / t / 1e12 / peak, 5)}
    ksize = {n.split("/")[0]: sh[0] for n, sh in eng.weight_specs() if n.endswith("kernel")}
    k3 = [s for s in slots if ksize.get(s[0]) == 3 and s[3] > 1e10]
    if k3:
        res["conv3x3_stack_frac"] = round(sum(s[3] for s in k3) / (sum(s[1] / s[2] for s in k3) * 1e-3) / 1e12 / peak, 5)
        res["conv3x3_stack_ms"] = round(sum(s[1] / s[2] for s in k3), 4)
    eng.close()
    return res


def leg_config5(torch, np, pseg_amd, synth, dev):
    """BASELINE.json configs[4]: 6-class predict at 4096x3072 + cc_majority vote + the four masks, everything resident
    in HBM, uint8 label map end to end.  GB/s = SURVEY.md 8d algorithmic bytes (vote: 2 reads + 1 write of the label map +
    1 read of the binarisation = 4 B/px; masks: 1 + 1 in, 12 out = 14 B/px) / time."""
    import ctypes
    from pseg_amd import engine as E
    H, W, C = 4096, 3072, 6
    eng = pseg_amd.Engine("fcn_skip", C, device=dev.index, mode=pseg_amd.MODE_BF16)
    eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
    img, binary, _ = synth.synth_page(1000, H, W, C)
    d_img = torch.from_numpy(img).to(dev)
    d_bin = torch.from_numpy(binary).to(dev)
    d_lab = torch.empty((H, W), dtype=torch.uint8, device=dev)
    lut = torch.from_numpy(np.array([[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255]], np.uint8)).to(dev)
    outs = [torch.empty((H, W, 3), dtype=torch.uint8, device=dev) for _ in range(4)]   # color, overlay, inverted, fg_color (lib/output.py:44-60)
    L = E.lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    vp = ctypes.c_void_p
    predict = lambda: eng.predict_device(d_img.data_ptr(), H, W, d_labels_u8=d_lab.data_ptr(), stream=st)
    vote = lambda: E._check(L.pseg_cc_vote_device_u8(dev.index, vp(d_lab.data_ptr()), vp(d_bin.data_ptr()), H, W, C, vp(st)))
    masks = lambda: E._check(L.pseg_masks_device_u8(dev.index, vp(d_lab.data_ptr()), vp(d_bin.data_ptr()), vp(lut.data_ptr()), C, H, W,
                                                    vp(outs[0].data_ptr()), vp(outs[1].data_ptr()), vp(outs[2].data_ptr()), vp(outs[3].data_ptr()), vp(st)))
    tp, tv, tm = _sync_time(torch, predict, 5), _sync_time(torch, vote, 5), _sync_time(torch, masks, 5)
    px = H * W
    res = {"predict_ms": round(tp * 1e3, 4), "cc_vote_ms": round(tv * 1e3, 4), "masks_ms": round(tm * 1e3, 4),
           "pipeline_ms": round((tp + tv + tm) * 1e3, 4), "pipeline_Mpixels_s": round(px / (tp + tv + tm) / 1e6, 1),
           "cc_vote_alg_GBs": round(px * 4 / tv / 1e9, 1), "cc_vote_frac_hbm": round(px * 4 / tv / 1e9 / HBM_PEAK_GBS, 4),
           "masks_alg_GBs": round(px * 14 / tm / 1e9, 1), "masks_frac_hbm": round(px * 14 / tm / 1e9 / HBM_PEAK_GBS, 4),
           "what": "4096x3072, 6 classes, fcn_skip bf16 + cc_majority vote + the four masks of generate_output_masks (12 B/px out), uint8 labels, HBM-resident"}
    eng.close()
    L.pseg_release_workspace(dev.index)
    return res


def leg_f32(torch, pseg_amd, synth, weights, d_img, H, W, C, dev, arch):
    """The float32 engine (PSEG_MODE_F32_EXACT: the bit-exact referee, the default of the drop-in Network, and the cost of a
    whole-page referee in the label-exact mode) on the same page and weights: ms/page and fraction of the 157.3 TFLOP/s
    float32 MFMA peak on the algorithmic FLOPs."""
    e32 = pseg_amd.Engine(arch, C, device=dev.index, mode=pseg_amd.MODE_F32_EXACT)
    e32.set_weights(weights)
    lab = torch.empty((H, W), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    fn = lambda: e32.predict_device(d_img.data_ptr(), H, W, d_labels_u8=lab.data_ptr(), stream=st)
    t = _sync_time(torch, fn, 5, warm=2)
    e32.timing_enable(True)
    e32.timing_reset()
    for _ in range(3):
        fn()
    torch.cuda.synchronize(dev)
    slots = [s for s in e32.timing() if s[2] > 0]
    e32.timing_enable(False)
    res = {"ms_per_page": round(t * 1e3, 4), "Mpixels_s": round(H * W / t / 1e6, 1),
           "whole_net_frac_f32_peak": round(e32.flops_per_pixel() * H * W / t / 1e12 / PEAK_TFLOPS["f32"], 5),
           "per_kernel_ms": {s[0]: round(s[1] / s[2], 5) for s in slots},
           "what": "PSEG_MODE_F32_EXACT, %dx%d, %s, uint8 labels left in HBM; peak 157.3 TFLOP/s (dense f32 MFMA)" % (H, W, arch)}
    e32.close()
    for other in ("unet", "res_unet"):           # the 3x3 stacks in float32 (what the drop-in Network costs for them by default)
        eo = pseg_amd.Engine(other, C, device=dev.index, mode=pseg_amd.MODE_F32_EXACT)
        eo.set_weights(synth.glorot_weights(eo.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
        fo = lambda: eo.predict_device(d_img.data_ptr(), H, W, d_labels_u8=lab.data_ptr(), stream=st)
        to = _sync_time(torch, fo, 3, warm=1)
        res[other] = {"ms_per_page": round(to * 1e3, 3), "whole_net_frac_f32_peak": round(eo.flops_per_pixel() * H * W / to / 1e12 / PEAK_TFLOPS["f32"], 5)}
        eo.close()
    return res


def leg_train(torch, np, pseg_amd, synth, H, W, C, dev, arch, steps=10, warm=2):
    """BASELINE.json configs[3] on one rank: float32 train step (forward, loss + metrics, backward, per-tensor clipnorm 1,
    Adam lr 1e-4) on synthetic 2048x1536 pages with synthetic masks, batch of one page as the reference
    (lib/network.py:235-241).  Fraction of the float32 MFMA peak on 3 x the forward FLOPs."""
    e = pseg_amd.Engine(arch, C, device=dev.index, mode=pseg_amd.MODE_F32_EXACT)
    e.set_weights(synth.glorot_weights(e.weight_specs(), seed=42))
    e.train_init(clipnorm=1.0)
    pages = [synth.synth_page(2000 + i, H, W, C) for i in range(2)]
    losses = []

    def step(i):
        img, _, mask = pages[i % len(pages)]
        m = e.train_forward_backward(img, mask)
        e.train_apply(1e-4, 1.0)
        losses.append(float(m[0]))

    for i in range(warm):
        step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        step(warm + i)
    torch.cuda.synchronize(dev)
    t = (time.perf_counter() - t0) / steps
    res = {"ms_per_step": round(t * 1e3, 3), "steps_per_s": round(1.0 / t, 2), "Mpixels_s": round(H * W / t / 1e6, 1),
           "frac_f32_peak": round(3.0 * e.flops_per_pixel() * H * W / t / 1e12 / PEAK_TFLOPS["f32"], 5),
           "loss_first": round(losses[0], 5), "loss_last": round(losses[-1], 5), "steps": steps + warm,
           "what": "configs[3] on one rank: %s %d-class train step on %dx%d synthetic pages / masks, batch of one page, "
                   "Adam lr 1e-4 + clipnorm 1, host uint8 page + mask in, 4 metrics out (PCIe included); "
                   "FLOPs = 3 x forward" % (arch, C, H, W)}
    e.close()
    return res


def leg_api_path(np, pseg_amd, synth, dev):
    """The drop-in chain a caller of the reference API gets (lib/predictor.py:32-54): Predictor.predict_masks on a
    4096x3072 6-class page with the cc_majority post-processor -- predict, vote and generate_output_masks stay on the device,
    label map and the four masks come down once."""
    from ocr4all_pixel_classifier.lib.predictor import Predictor
    from ocr4all_pixel_classifier.lib.predictor_data import PredictSettings
    from ocr4all_pixel_classifier.lib.dataset import SingleData
    from ocr4all_pixel_classifier.lib.network import Network
    from ocr4all_pixel_classifier.lib.postprocess import vote_connected_component_class
    H, W, C = 4096, 3072, 6
    img, binary, _ = synth.synth_page(1000, H, W, C)
    net = Network("Predict", n_classes=C, exact=False, device=dev.index)
    net.model.set_weights(synth.glorot_weights(net.model.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
    from ocr4all_pixel_classifier.lib.colors import ColorMap
    cm = ColorMap({(255, 255, 255): (0, "bg"), (255, 0, 0): (1, "a"), (0, 255, 0): (2, "b"), (0, 0, 255): (3, "c"),
                   (255, 255, 0): (4, "d"), (0, 255, 255): (5, "e")})
    settings = PredictSettings(n_classes=C, network=None, output=None, color_map=cm, post_process=[vote_connected_component_class])
    pred = Predictor(settings, network=net)
    data = SingleData(image=img, binary=binary, original_shape=img.shape, image_path="p.png")
    pred.predict_masks(data)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        masks = pred.predict_masks(data)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    t = ts[len(ts) // 2]
    return {"predict_masks_ms": round(t * 1e3, 3), "Mpixels_s": round(H * W / t / 1e6, 1),
            "what": "Predictor.predict_masks (lib/predictor.py:44-54) at 4096x3072x6 with cc_majority, NumPy page in, the four "
                    "(H,W,3) masks out as NumPy arrays (37.7 MB each over PCIe); median of 5"}


def leg_cpu_baseline(np, weights, page, arch):
    """the float32 restatement of the predict path on torch-CPU (oneDNN convolutions), 3 warm-ups
    + median of 10 for both, at n = all cores on the full page and at n = 1 on a 512x512 crop (configs[0]'s size) so that
    the whole leg stays within ~30 s of CPU work.  The C oracle (sequential-fmaf port) is timed once on 256 rows."""
    import torch
    import oracle
    from oracle import torch_cpu
    ncores = os.cpu_count() or 1
    try:
        ncores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:   # a container's CPU share (cgroup quota) is what the process really gets, whatever the affinity mask says
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            ncores = max(1, min(ncores, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    H, W = page.shape
    # threads for the "all cores" leg: the visible cores, or fewer when the host hands this process a smaller share than
    # it shows (256 threads on a 16-core share ran 100x slower than one thread): probe a few counts on a 512x512 crop
    crop = page[:512, :512].copy()
    best_n, probe = ncores, None
    for n in sorted({ncores, min(ncores, 64), min(ncores, 32), min(ncores, 16)}, reverse=True):
        torch_cpu.fcn_forward(arch, weights, crop, threads=n)
        t0 = time.perf_counter()
        torch_cpu.fcn_forward(arch, weights, crop, threads=n)
        tn = time.perf_counter() - t0
        if probe is None or tn < probe:
            best_n, probe = n, tn
    ncores_seen, ncores = ncores, best_n
    big = page if probe * (H * W) / (512 * 512) < 2.0 else page[:1024, :768].copy()
    t_all = torch_cpu.time_predict(arch, weights, np.ascontiguousarray(big), ncores, warmup=3, reps=10)
    small = np.ascontiguousarray(page[:512, :512])
    t_one = torch_cpu.time_predict(arch, weights, small, 1, warmup=3, reps=10)
    torch.set_num_threads(ncores)
    oracle.build()
    rows = np.ascontiguousarray(page[:256])
    oracle.forward(arch, weights, rows[:64, :64].copy(), "f32")
    t0 = time.perf_counter()
    oracle.forward(arch, weights, rows, "f32")
    t_port = time.perf_counter() - t0
    return {"value": round(big.size / t_all / 1e6, 3), "unit": "Mpixels/s", "cores": ncores, "kind": "port",
            "impl": "restatement (torch-CPU/oneDNN, float32); the reference's TensorFlow-CPU path cannot run offline",
            "sample": "%dx%d page, n=%d threads (fastest of the probed counts; %d cores visible), 3 warm-ups + median of 10 (%.3f s per page)"
                      % (big.shape[0], big.shape[1], ncores, ncores_seen, t_all),
            "n1": {"value": round(small.size / t_one / 1e6, 4), "cores": 1,
                   "sample": "512x512 crop (configs[0] size), 3 warm-ups + median of 10 (%.2f s per page)" % t_one},
            "oracle_port": {"value": round(rows.size / t_port / 1e6, 4), "cores": oracle.num_threads(),
                            "sample": "rows 0..256 of the page, oracle/pseg_oracle.c (sequential fmaf chains, OpenMP), one pass %.2f s" % t_port}}


# ---------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------
def run_rank(args):
    import numpy as np
    import torch
    import pseg_amd
    from pseg_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # rehearsal knobs (not used by the driver): PSEG_BENCH_BACKEND=gloo and PSEG_BENCH_ONE_GPU=1 run the
    # N > 1 code path with several ranks on a one-GPU box (gloo barrier / max-reduce, every rank on cuda:0)
    backend = os.environ.get("PSEG_BENCH_BACKEND", "nccl")
    if os.environ.get("PSEG_BENCH_ONE_GPU"):
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        if torch.cuda.device_count() <= local_rank:
            sys.stderr.write("bench.py: rank %d needs cuda:%d but only %d device(s) are visible\n" % (rank, local_rank, torch.cuda.device_count()))
            return 3
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180),
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    else:
        dist = None
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    H, W, C = args.height, args.width, args.classes
    mode = pseg_amd.MODE_BF16 if args.mode == "bf16" else pseg_amd.MODE_F32_EXACT
    eng = pseg_amd.Engine(args.arch, C, device=local_rank, mode=mode)
    weights = synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05)
    eng.set_weights(weights)

    # synthetic pages, resident in HBM before the timed region (page index = global page id)
    host_pages = []
    for p in range(args.pages):
        img, _, _ = synth.synth_page(rank * args.pages + p, H, W, C)
        host_pages.append(img)
    # one behind the other, as pseg_predict_pages_device takes them (lib/predictor.py:27-30's page loop as one call)
    d_pages = torch.from_numpy(np.stack(host_pages)).to(dev)
    d_labels = torch.empty((args.pages, H, W), dtype=torch.uint8, device=dev)
    pages = [d_pages[p] for p in range(args.pages)]
    labels = [d_labels[p] for p in range(args.pages)]
    # the hot path is launched on a stream of this process's own (torch's default stream has the raw handle 0, which
    # the library reads as "the engine's stream"): the per-step torch events below are recorded on the same stream
    tstream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream

    def step():
        if args.pages > 1 and not args.page_by_page:
            eng.predict_pages_device(d_pages.data_ptr(), args.pages, H, W, d_labels_u8=d_labels.data_ptr(), stream=stream)
            return
        for img_t, lab_t in zip(pages, labels):
            eng.predict_device(img_t.data_ptr(), H, W, d_labels_u8=lab_t.data_ptr(), stream=stream)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Order of the legs: the per-step spread pass and the per-kernel roofline pass run FIRST, then the W warm-up steps, then the
    # K timed steps.  A cold MI355X needs about 25 pages (12 ms) of load before it holds its sustained clocks: with W = 5
    # straight after process start the 20-step region measured 0.509 ms/page, with W = 50 or K = 200 0.457-0.461 ms/page
    # (gpurun, round 2) -- the sustained rate is the one a page stream sees, and the one the roofline leg is priced at.
    for _ in range(32):      # spin-up, untimed: every leg below (spread, roofline, the timed region) sees the sustained clocks
        step()
    torch.cuda.synchronize(dev)
    # per-step spread (separate pass, events on the launch stream): the timed region below is a single sample
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(args.steps, 10))]
    for a, b in evs:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize(dev)
    per = sorted(a.elapsed_time(b) for a, b in evs)
    step_stats = {"ms_per_step_median": round(per[len(per) // 2], 4), "ms_per_step_min": round(per[0], 4),
                  "ms_per_step_max": round(per[-1], 4), "samples": len(per)}

    # ---- roofline: per-kernel HIP-event timing on the same stream, separate untimed pass --------
    roof = None
    eng.timing_enable(True)
    eng.timing_reset()
    nroof = max(3, min(args.steps, 10))
    for _ in range(nroof):
        step()
    torch.cuda.synchronize(dev)
    slots = [s for s in eng.timing() if s[2] > 0]
    eng.timing_enable(False)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    eng.status(stream)      # (outside the timed region) a kernel-side error of any step above is an error of the bench
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    total_px = float(world) * args.pages * H * W * args.steps
    value = total_px / dt / 1e6

    if slots:
        # a launch of the page entry covers several pages (page slots): every figure below is per PAGE -- a slot's time over the
        # pages of the pass -- so that it compares with the slot's algorithmic FLOPs of one page
        paged = args.pages > 1 and not args.page_by_page
        per_page = (lambda s: s[1] / (nroof * args.pages)) if paged else (lambda s: s[1] / s[2])
        name, ms, n, flops = max(slots, key=lambda s: s[1])
        avg_ms = per_page((name, ms, n, flops))
        achieved = flops / (avg_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.mode]
        total_ms = sum(per_page(s) for s in slots)
        # HBM traffic of the dominant kernel: rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate
        # runs, gfx950 correction applied) recorded under profiles/ by tools/pmc_traffic.py -- replayed, not measured here
        traffic, traffic_src = None, None
        for fn in ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01e_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", fn)) as f:
                    tr = json.load(f).get(name)
                if tr and (H, W, C, args.arch, args.mode) == (2048, 1536, 3, "fcn_skip", "bf16"):
                    traffic, traffic_src = tr["hbm_read_bytes"] + tr["hbm_write_bytes"], fn
                    break
            except (OSError, ValueError, KeyError):
                continue
        roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 3), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 5), "traffic": traffic,
                "traffic_unit": "HBM bytes per launch, replayed from profiles/%s (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes)" % traffic_src,
                "avg_ms": round(avg_ms, 5), "launches": int(n),
                "flop_per_launch": flops,
                "whole_net_frac": round(eng.flops_per_pixel() * H * W / (dt / args.steps / args.pages) / 1e12 / peak, 5),
                "whole_net_frac_what": "algorithmic FLOPs of the page / ms_per_step (the timed region); per_kernel_ms is a separate pass "
                                       "with event timing on, whose sum prices whole_net_frac_kernel_sum",
                "whole_net_frac_kernel_sum": round(eng.flops_per_pixel() * H * W / (total_ms * 1e-3) / 1e12 / peak, 5),
                "per_kernel_ms": {s[0]: round(per_page(s), 5) for s in slots}}
        if args.mode == "bf16":
            # context, not the contract's `peak`: what the chip sustains on this instruction stream (tools/microtests/kloop_tiles.hip, round 5)
            roof["measured_ceiling"] = {"TFLOP/s": 1596.0, "frac_of_it": round(achieved / 1596.0, 5),
                                        "what": "bare v_mfma_f32_16x16x32_bf16 loop, two waves per SIMD, random operands, every CU busy: the clock the chip "
                                                "holds under that load gives 0.64 of the nominal 2 500 (profiles/r05_kloop_tiles_microbench.txt); `achieved` is "
                                                "algorithmic work -- the dominant launch issues 1.30 x that (K 500 -> 520, Cout 30 -> 32, conv1 on the halo)"}
        if paged:
            roof["per_kernel_ms_what"] = "per page: the layer's time over the pages of the pass (launches of the low-resolution layers cover a unit of pages)"
        ksize = {n.split("/")[0]: sh[0] for n, sh in eng.weight_specs() if n.endswith("kernel")}
        k3 = [s for s in slots if ksize.get(s[0]) == 3 and s[3] > 1e10]
        if k3:
            roof["conv3x3_stack_frac"] = round(sum(s[3] for s in k3) / (sum(per_page(s) for s in k3) * 1e-3) / 1e12 / peak, 5)

    extra = dict(step_stats)
    cpu = None
    if rank == 0 and world == 1:
        default_cfg = (H, W, C, args.arch, args.mode) == (2048, 1536, 3, "fcn_skip", "bf16")
        if not args.no_extra and default_cfg:
            for key, fn in (("pages32", lambda: leg_pages32(torch, np, eng, synth, H, W, C, dev, args.steps, args.warmup)),
                            ("host_path", lambda: leg_host_path(np, pseg_amd, eng, synth, H, W, C)),
                            ("f32", lambda: leg_f32(torch, pseg_amd, synth, weights, pages[0], H, W, C, dev, args.arch)),
                            ("train", lambda: leg_train(torch, np, pseg_amd, synth, H, W, C, dev, args.arch)),
                            ("label_exact", lambda: leg_label_exact(torch, np, pseg_amd, eng, pages[0], H, W, dev, synth, C, args.arch)),
                            ("unet", lambda: leg_arch(torch, pseg_amd, synth, "unet", H, W, C, dev)),
                            ("res_unet", lambda: leg_arch(torch, pseg_amd, synth, "res_unet", H, W, C, dev)),
                            ("config5", lambda: leg_config5(torch, np, pseg_amd, synth, dev)),
                            ("api_path", lambda: leg_api_path(np, pseg_amd, synth, dev))):
                try:
                    extra[key] = fn()
                except Exception as ex:   # an extra leg must not take the headline line down with it
                    extra[key] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if not args.no_cpu_baseline and args.arch in ("fcn", "fcn_skip"):
            try:
                cpu = leg_cpu_baseline(np, weights, host_pages[0], args.arch)
            except Exception as ex:
                cpu = {"error": "%s: %s" % (type(ex).__name__, ex)}

    if rank == 0:
        out = {
            "metric": "Mpixels/s classified (%dx%d, %d-class)" % (H, W, C),
            "value": round(value, 3),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.mode,
            "data": "synthetic pages (numpy default_rng(1000+i)), glorot random-init weights (default_rng(42))",
            "config": {"workload": (("configs[1]: single %dx%d page, %d-class %s predict" % (H, W, C, args.arch)) if world == 1 and args.pages == 1 else
                                    ("configs[2] shape: %d independent %dx%d pages per rank per step, %d-class %s predict, page-parallel, "
                                     "no data-path collective, %s" % (args.pages, H, W, C, args.arch,
                                     "one pseg_predict_device call per page" if args.page_by_page else "one pseg_predict_pages_device call per step (page slots: the low-resolution layers take all pages of a unit in one launch)")))
                                   + ", inputs resident in HBM, uint8 label maps left in HBM (value = HBM-resident rate per the measurement "
                                     "contract; SURVEY 8d's pinned-host-in / host-out rate is value_host_path)",
                       "pages_per_rank_per_step": args.pages, "parallelism": "page-parallel x%d" % world,
                       "per_rank_baseline": ("extra.pages32 of the --gpus 1 line (the same %d pages per step through pseg_predict_pages_device on one GPU), "
                                             "not its `value` (one page per step)" % args.pages) if world > 1 else None},
            "roofline": roof,
            "cpu_baseline": cpu,
            "extra": extra,
        }
        hp = extra.get("host_path") if isinstance(extra.get("host_path"), dict) else None
        if hp and "uint8" in hp:
            # SURVEY.md 8d's boundary metric, named at the top level next to the HBM-resident `value`
            out["value_host_path"] = {"value": hp["uint8"]["Mpixels_s"], "unit": "Mpixels/s", "ms_per_page": hp["uint8"]["ms_per_page"],
                                      "what": "SURVEY 8d: uint8 pages in pinned host memory -> uint8 label maps in pinned host memory, "
                                              "pseg_predict_batch (PCIe both ways, overlapped with compute); int64 labels: extra.host_path.int64"}
            extra["hbm_resident"] = {"Mpixels_s": out["value"], "ms_per_page": round(dt / args.steps / args.pages * 1e3, 4)}
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()
    return 0


def leg_label_exact(torch, np, pseg_amd, eng, d_img, H, W, dev, synth=None, C=3, arch="fcn_skip"):
    """Label-exact throughput mode: bf16 pass + margin map, float32 referee on the tiles that hold near-ties; the label
    map equals the float32 engine's.  Reports the re-evaluated tile fraction and the per-page cost -- for the bench's
    random-init weights (near-ties everywhere: the referee takes the whole page, the mode's worst case) and, under
    "trained", for weights after 150 Adam steps on synthetic pages (confident regions keep their bf16 labels)."""
    if not hasattr(eng, "predict_exact_labels_device"):
        return {"error": "not built"}
    st = torch.cuda.current_stream(dev).cuda_stream

    def measure(e, img_t):
        lab = torch.empty((H, W), dtype=torch.uint8, device=dev)
        e.predict_exact_labels_device(img_t.data_ptr(), H, W, lab.data_ptr(), stream=st)
        torch.cuda.synchronize(dev)
        t = _sync_time(torch, lambda: e.predict_exact_labels_device(img_t.data_ptr(), H, W, lab.data_ptr(), stream=st), 5, warm=1)
        info = e.label_exact_stats()
        info["ms_with_referee"] = round(t * 1e3, 4)
        info["Mpixels_s"] = round(H * W / t / 1e6, 1)
        return info, lab

    info, _ = measure(eng, d_img)
    info["weights"] = "random init (the bench's): worst case, whole-page referee"
    if synth is not None:
        try:
            e32 = pseg_amd.Engine(arch, C, mode=pseg_amd.MODE_F32_EXACT)
            e32.set_weights(synth.glorot_weights(e32.weight_specs(), seed=7))
            e32.train_init(clipnorm=1.0)
            tp = [synth.synth_page(s, 128, 160, C) for s in range(6)]
            first = last = None
            for it in range(150):
                img, _, mask = tp[it % len(tp)]
                last = e32.train_forward_backward(img, mask)[0]
                e32.train_apply(2e-3)
                first = last if first is None else first
            eb = pseg_amd.Engine(arch, C, mode=pseg_amd.MODE_BF16)
            eb.set_weights(e32.get_weights())
            page = torch.from_numpy(synth.synth_page(99, H, W, C)[0]).to(dev)
            tr, lab = measure(eb, page)
            l32 = torch.empty((H, W), dtype=torch.uint8, device=dev)
            e32.predict_device(page.data_ptr(), H, W, d_labels_u8=l32.data_ptr(), stream=st)
            torch.cuda.synchronize(dev)
            tr["equal_to_float32_labels"] = bool(torch.equal(lab, l32))
            tb = _sync_time(torch, lambda: eb.predict_device(page.data_ptr(), H, W, d_labels_u8=lab.data_ptr(), stream=st), 10, warm=2)
            tr["ms_bf16_only"] = round(tb * 1e3, 4)
            tr["weights"] = "150 Adam steps (lr 2e-3) on six 128x160 synthetic pages, loss %.3f -> %.3f" % (first, last)
            tr["page"] = "synthetic text page (class boundaries through nearly every 32-px block at the 48-px line pitch)"
            info["trained"] = tr
            # a page with large single-class areas (content in the top-left quarter of the width and height, paper elsewhere:
            # title pages, chapter ends): the referee works in parts
            rng = np.random.default_rng(7)
            sp = (255 - np.clip(rng.normal(225.0, 8.0, size=(H, W)), 0, 255).astype(np.uint8)).astype(np.uint8)
            sp[:H // 4 // 32 * 32, :W // 4 // 32 * 32] = synth.synth_page(7, H, W, C)[0][:H // 4 // 32 * 32, :W // 4 // 32 * 32]
            spage = torch.from_numpy(sp).to(dev)
            e32b = pseg_amd.Engine(arch, C, mode=pseg_amd.MODE_F32_EXACT)
            e32b.set_weights(synth.glorot_weights(e32b.weight_specs(), seed=7))
            e32b.train_init(clipnorm=1.0)
            for it in range(200):
                img, _, mask = tp[it % len(tp)]
                e32b.train_forward_backward(img, mask)
                e32b.train_apply(1e-3)
            eb.set_weights(e32b.get_weights())
            sr, slab = measure(eb, spage)
            e32b.predict_device(spage.data_ptr(), H, W, d_labels_u8=l32.data_ptr(), stream=st)
            torch.cuda.synchronize(dev)
            sr["equal_to_float32_labels"] = bool(torch.equal(slab, l32))
            sr["weights"] = "200 Adam steps (lr 1e-3) on the same six pages"
            sr["page"] = "content in the top-left quarter of the page's width and height, paper elsewhere"
            info["sparse_page"] = sr
            e32b.close()
            eb.close()
            e32.close()
        except Exception as ex:   # noqa: BLE001 -- an extra leg never takes the bench line down
            info["trained"] = {"error": repr(ex)}
    return info


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s\n" % (args.gpus, os.environ.get("WORLD_SIZE")))
        return 2
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
