"""Library GEMM (hipBLASLt through torch.matmul, bf16, fp32 accumulate) at the implicit-GEMM shapes of the UNet-VGG convolutions,
random data, kept running ~40 ms before timing: the rate a tuned dense GEMM reaches on this part WITHOUT gathering the 3x3 patches."""
import torch
N = 32
LAYERS = [("conv1.3", 384, 64, 64), ("conv2.10", 192, 128, 128), ("conv3.17", 96, 256, 256), ("conv4.27", 48, 512, 512), ("conv5.34", 24, 512, 512),
          ("dec4.0", 48, 768, 512), ("dec3.0", 96, 512, 256), ("dec2.0", 192, 192, 128), ("dec1.0", 384, 96, 32)]
for name, H, Cin, Cout in LAYERS:
    M, K = N * H * H, 9 * Cin
    if M * K * 2 > 12e9:
        M //= 4          # keep the explicit patch matrix under 12 GB (the rate does not depend on M at this size)
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = (torch.randn(Cout, K, device="cuda") * 0.05).to(torch.bfloat16)
    f = lambda: torch.matmul(a, b.t())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    reps = max(5, int(40 / max(e0.elapsed_time(e1), 1e-3)))
    for _ in range(reps): f()
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:9s} M={M:8d} N={Cout:4d} K={K:5d}  {ms*1e3:8.1f} us  {2.0*M*K*Cout/ms/1e9:6.0f} TF", flush=True)
    del a, b
