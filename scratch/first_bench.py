"""First-layer kernels alone (batch 32, 3 x 384 x 384 -> 64 channels, bf16)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-segmentation_amd"))
from segk import ops
from segk.ops import Act
N, H, Cin, Cout = 32, 384, int(os.environ.get("CIN", 3)), 64
x = torch.randn(N, Cin, H, H, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.2
b = torch.randn(Cout, device="cuda")
y = Act.empty(N, H, H, Cout, torch.bfloat16, "cuda")
dy = Act(torch.randn(N, H, H, Cout, device="cuda").bfloat16())
def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
mb = N * H * H * Cout * 2 / 1e6
t = timeit(lambda: ops.conv_first_fwd(x, w, b, y, want_stats=True))
print(f"conv_first_fwd (with statistics): {t:.1f} us, {mb / t * 1e-6 * 1e6 / 1e3:.2f} TB/s of output")
t = timeit(lambda: ops.conv_first_fwd(x, w, b, y, want_stats=False))
print(f"conv_first_fwd (no statistics): {t:.1f} us")
t = timeit(lambda: ops.im2col3x3(x, torch.bfloat16))
print(f"im2col3x3: {t:.1f} us")
if hasattr(ops, "conv_first_wgrad"):
    dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
    t = timeit(lambda: ops.conv_first_wgrad(x, dy, dw))
    print(f"conv_first_wgrad: {t:.1f} us, {mb / t * 1e-6 * 1e6 / 1e3:.2f} TB/s of dy")
p = ops.im2col3x3(x, torch.bfloat16)
dw2 = torch.empty(Cout, 9 * Cin, device="cuda")
t = timeit(lambda: ops.conv_wgrad(p, dy, dw2, taps=1, cin_real=9 * Cin))
print(f"patch-image wgrad: {t:.1f} us")
