"""Probe: library bf16 GEMM time for the four unet up-conv layers in the split form
(D[M][4*Cout] = src[M][Cin] @ W[4*Cout][Cin]^T), M = source pixels at 2048x1536."""
import torch, time
torch.cuda.is_available()
shapes = [("conv2d_10", 128 * 96, 4 * 512, 1024), ("conv2d_13", 256 * 192, 4 * 256, 512),
          ("conv2d_16", 512 * 384, 4 * 128, 256), ("conv2d_19", 1024 * 768, 4 * 64, 128)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): d = a @ w.t()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): d = a @ w.t()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%s M=%d N=%d K=%d  %.1f us  %.0f TFLOP/s  (D %.0f MB)" % (name, M, N, K, dt * 1e6, 2.0 * M * N * K / dt / 1e12, M * N * 2 / 1e6))
