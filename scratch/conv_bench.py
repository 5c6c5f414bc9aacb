"""Micro-benchmark of the MFMA convolution kernels on the UNet-VGG layer shapes (batch 32, bf16)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-segmentation_amd"))
from segk import ops
from segk.ops import Act

N = int(os.environ.get("N", 32))
dtype = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32
# (name, H, Cin, Cout)
LAYERS = [("conv1.3", 384, 64, 64), ("conv2.7", 192, 64, 128), ("conv2.10", 192, 128, 128), ("conv3.14", 96, 128, 256),
          ("conv3.17", 96, 256, 256), ("conv4.24", 48, 256, 512), ("conv4.27", 48, 512, 512), ("conv5.34", 24, 512, 512),
          ("center.0", 12, 512, 512), ("center.2", 12, 512, 256), ("dec5.0", 24, 768, 512), ("dec5.2", 24, 512, 256),
          ("dec4.0", 48, 768, 512), ("dec4.2", 48, 512, 256), ("dec3.0", 96, 512, 256), ("dec3.2", 96, 256, 64),
          ("dec2.0", 192, 192, 128), ("dec2.2", 192, 128, 32), ("dec1.0", 384, 96, 32)]
if os.environ.get("SET") == "unetpp":     # UNet++ (batch 16): the full-resolution nested decoder and a few inner nodes
    N = int(os.environ.get("N", 16))
    LAYERS = [("x0_1.c1", 384, 128, 64), ("x0_2.c1", 384, 192, 64), ("x0_3.c1", 384, 256, 64), ("x0_4.c1", 384, 320, 64),
              ("x0_j.c2", 384, 64, 64), ("x1_1.c1", 192, 256, 128), ("x1_3.c1", 192, 512, 128), ("x2_2.c1", 96, 768, 256),
              ("x3_1.c1", 48, 1024, 512), ("x4_0.c2", 24, 1024, 1024)]
only = os.environ.get("ONLY")
if only:
    LAYERS = [l for l in LAYERS if l[0] in only.split(",")]
which = sys.argv[1:] or ["fwd", "dgrad", "wgrad"]
tot = {k: 0.0 for k in which}
totf = 0.0
for name, H, Cin, Cout in LAYERS:
    torch.manual_seed(0)
    x = Act(torch.randn(N, H, H, Cin, device="cuda").to(dtype))
    dy = Act(torch.randn(N, H, H, Cout, device="cuda").to(dtype))
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    wf, wb = ops.conv_pack(w, dtype)
    y = Act.empty(N, H, H, Cout, dtype, "cuda")
    dx = Act.empty(N, H, H, Cin, dtype, "cuda")
    dw = torch.empty_like(w)
    fl = 2.0 * N * H * H * Cin * Cout * 9
    part = ops.conv_stat_partials(y, fine=bool(os.environ.get("FINE")))
    bias = torch.randn(Cout, device="cuda") if os.environ.get("BIAS") else None
    fns = {"fwd": lambda: ops.conv_fwd(x, wf, Cout, y, bias=bias), "fwds": lambda: ops.conv_fwd(x, wf, Cout, y, bias=bias, stat_partials=part), "dgrad": lambda: ops.conv_fwd(dy, wb, Cin, dx, mask=x if os.environ.get("MASK") else None),
           "wgrad": lambda: ops.conv_wgrad(x, dy, dw)}
    line = f"{name:9s} {H:3d}^2 {Cin:3d}->{Cout:3d} {fl/1e9:7.1f} GF |"
    for k in which:
        f = fns[k]
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # SUSTAIN=ms: keep the kernel running that long before and during the measurement, so the figure is taken at the
        # clock the chip holds under load and not in the boost window after an idle phase
        reps = 5
        sustain = float(os.environ.get("SUSTAIN", 0))
        if sustain:
            e0.record(); f(); e1.record(); torch.cuda.synchronize()
            reps = max(5, int(sustain / max(e0.elapsed_time(e1), 1e-3)))
            for _ in range(reps): f()
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tot[k] += ms
        line += f" {k} {ms*1e3:7.1f} us {fl/ms/1e9:6.0f} TF |"
    totf += fl
    if os.environ.get("CHECK"):      # checksums of the results (compare two builds / modes on identical inputs)
        torch.cuda.synchronize()
        bits = lambda t: int((t.view(torch.int16).long() * (torch.arange(t.numel(), device=t.device).view(t.shape) % 251 + 1)).sum())
        line += (f" sum|y| {float(y.buf.float().abs().sum()):.6e} sum|dx| {float(dx.buf.float().abs().sum()):.6e} sum|dw| {float(dw.abs().sum()):.6e}"
                 f" bits(y) {bits(y.buf)} bits(dx) {bits(dx.buf)}")
        if "fwds" in which:
            line += f" stat sums {float(part[:, 0].double().sum()):.9e} {float(part[:, 1].double().sum()):.9e} {float(part[:, 1].double().sum(0).max()):.9e}"
    print(line, flush=True)
print("TOTAL", {k: f"{v:.2f} ms {totf/v/1e9:.0f} TF" for k, v in tot.items()})
