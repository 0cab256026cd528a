"""Calibration only (not used by the library): what the vendor GEMM (hipBLASLt behind torch.matmul) reaches on the four
b = 256 linears of ViT-L/14, bf16 in / bf16 out, random operands, to read gemm_bench's numbers against."""
import torch, time
M = 256 * 257
shapes = {"qkv": (3072, 1024), "out": (1024, 1024), "fc1": (4096, 1024), "fc2": (1024, 4096)}
dev = "cuda:0"
for name, (N, K) in shapes.items():
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    for fn, label in ((lambda: torch.matmul(x, w.t()), "matmul"), (lambda: torch.nn.functional.linear(x, w, b), "linear+bias")):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        print(f"{name:4s} M={M} N={N} K={K} {label:12s}: {ms*1e3:7.1f} us  {2.0*M*N*K/ms*1e-9:7.0f} TFLOP/s", flush=True)
