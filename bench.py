#!/usr/bin/env python3
"""bench.py -- Mpixels/s classified on synthetic 2048x1536 3-class pages (BASELINE.json).

One "step" = one pass of the predict hot path (x/255 -> pad -> fcn_skip -> crop -> logits ->
argmax) over one batch of `--pages` synthetic pages per rank, inputs already resident in HBM,
label maps left in HBM.  N=1 runs BASELINE.json configs[1] (single 2048x1536 page, 3 classes,
bf16 activations).  N>1: one process per GPU, independent pages per rank, no data-path
collective (weak scaling); value = pixels of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant
kernel, algorithmic FLOPs / HIP-event duration measured here) and `cpu_baseline` (the CPU
oracle timed on this host's cores; reference TensorFlow path is not runnable offline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "page-segmentation_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md: dense MFMA peaks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=("bf16", "f32"), default="bf16")
    ap.add_argument("--arch", default="fcn_skip")
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--height", type=int, default=2048)
    ap.add_argument("--width", type=int, default=1536)
    ap.add_argument("--pages", type=int, default=1, help="pages per rank per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=2048,
                    help="rows of the page the CPU oracle is timed on")
    args = ap.parse_args()

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
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
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
    pages = []
    for p in range(args.pages):
        img, _, _ = synth.synth_page(rank * args.pages + p, H, W, C)
        pages.append(torch.from_numpy(img).to(dev))
    labels = [torch.empty((H, W), dtype=torch.uint8, device=dev) for _ in range(args.pages)]
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        for img_t, lab_t in zip(pages, labels):
            eng.predict_device(img_t.data_ptr(), H, W, d_labels_u8=lab_t.data_ptr(), stream=stream)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    total_px = float(world) * args.pages * H * W * args.steps
    value = total_px / dt / 1e6

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
    if slots:
        name, ms, n, flops = max(slots, key=lambda s: s[1])
        avg_ms = ms / n
        achieved = flops / (avg_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.mode]
        total_ms = sum(s[1] / s[2] for s in slots)
        # HBM traffic of the dominant kernel: rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate
        # runs, gfx950 correction applied) recorded under profiles/ -- see profiles/r01e_traffic.json
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01e_traffic.json")) as f:
                tr = json.load(f).get(name)
            if tr and (H, W, C, args.arch, args.mode) == (2048, 1536, 3, "fcn_skip", "bf16"):
                traffic = tr["hbm_read_bytes"] + tr["hbm_write_bytes"]
        except (OSError, ValueError, KeyError):
            traffic = None
        roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 3), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 5), "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01e_traffic.json)",
                "avg_ms": round(avg_ms, 5), "launches": int(n),
                "flop_per_launch": flops,
                "whole_net_frac": round(eng.flops_per_pixel() * H * W / (total_ms * 1e-3) / 1e12 / peak, 5),
                "per_kernel_ms": {s[0]: round(s[1] / s[2], 5) for s in slots}}
        # the 3x3 conv stack on its own (north_star names it for unet): layers whose kernel is 3x3, without the
        # Cin = 1 first layer (a write stream, not MFMA work)
        ksize = {n.split("/")[0]: sh[0] for n, sh in eng.weight_specs() if n.endswith("kernel")}
        k3 = [s for s in slots if ksize.get(s[0]) == 3 and s[3] > 1e10]
        if k3:
            roof["conv3x3_stack_frac"] = round(sum(s[3] for s in k3) / (sum(s[1] / s[2] for s in k3) * 1e-3) / 1e12 / peak, 5)

    # ---- CPU baseline: the oracle (port of the reference semantics) on this host ---------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        oracle.build()
        rows = min(H, args.cpu_sample_rows)
        rows -= rows % 32
        img0 = pages[0][:max(rows, 32)].cpu().numpy()
        ow = {k: v for k, v in weights.items()}
        oracle.forward(args.arch, ow, img0[:64, :64].copy(), "f32")        # warm the library
        t0 = time.perf_counter()
        z = oracle.forward(args.arch, ow, np.ascontiguousarray(img0), "f32")
        np.argmax(z, -1)
        tc = time.perf_counter() - t0
        cpu = {"value": round(img0.shape[0] * img0.shape[1] / tc / 1e6, 4), "unit": "Mpixels/s",
               "cores": oracle.num_threads(), "kind": "port",
               "sample": "rows 0..%d of page 0 (%dx%d px), f32 oracle (OpenMP), one pass, %.1f s; the "
                         "reference's TensorFlow-CPU path cannot run offline" % (img0.shape[0], img0.shape[0], img0.shape[1], tc)}

    if rank == 0:
        out = {
            "metric": "Mpixels/s classified (2048x1536, 3-class)",
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
            "config": {"workload": "configs[1]: single %dx%d page, %d-class %s predict, inputs resident in HBM"
                                   % (H, W, C, args.arch),
                       "pages_per_rank_per_step": args.pages, "parallelism": "page-parallel x%d" % world},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
