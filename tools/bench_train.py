"""Train-step timing (BASELINE.json configs[4] shape: 3-class fcn_skip, synthetic masks, Adam lr 1e-4,
clipnorm 1).  One rank per GPU; with WORLD_SIZE > 1 the flat gradient is all-reduced over RCCL.
Prints one JSON line on rank 0.  Not the headline metric (bench.py is)."""
import argparse, json, os, sys, time
os.environ.setdefault("PSEG_PLAN_FROM_ENV", "1")   # PSEG_* of the environment -> plan switches of the engines created here
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "page-segmentation_amd")]
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--arch", default="fcn_skip")
    ap.add_argument("--pages", type=int, default=2, help="synthetic pages each rank cycles through (configs[3]: 8)")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--log-every", type=int, default=0, help="record the mean loss / accuracy of every block of this many steps")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.is_available()
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world)
    from pseg_amd import engine as E, synth
    from pseg_amd.parallel import allreduce_gradients
    eng = E.Engine(a.arch, 3, device=local, mode=E.MODE_F32_EXACT)
    eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42))
    eng.train_init(clipnorm=1.0)
    pages = [synth.synth_page(1000 + rank * 8 + i, a.height, a.width, 3) for i in range(a.pages)]
    curve, blk = [], []

    def step(i):
        img, _, mask = pages[i % len(pages)]
        m = eng.train_forward_backward(img, mask)
        allreduce_gradients(eng, world)
        eng.train_apply(a.lr, 1.0 / world)
        return m

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        m = step(i)
        if a.log_every:
            blk.append(m[:2])
            if len(blk) == a.log_every:
                curve.append([i + 1] + [round(float(v), 5) for v in np.mean(np.asarray(blk), 0)])
                blk = []
                if rank == 0:
                    print("step %d loss %.5f acc %.5f" % tuple(curve[-1]), file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        px = a.height * a.width
        print(json.dumps({"metric": "train steps/s (batch of one page per rank)", "value": world * a.steps / dt,
                          "ms_per_step": 1e3 * dt / a.steps, "n_gpus": world, "page": [a.height, a.width],
                          "Mpx_per_s": world * a.steps * px / dt / 1e6, "dtype": "f32", "last_metrics": list(map(float, m)),
                          "arch": a.arch, "lr": a.lr, "pages_per_rank": a.pages, "steps": a.steps,
                          "loss_curve_[step,loss,accuracy]": curve}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
