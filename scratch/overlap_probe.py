"""How well do an MFMA-bound convolution and an HBM-bound streaming kernel share the GPU when they run on two streams?
Prints the time of K convolutions alone, K norm_apply passes alone, and both sets launched concurrently."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-segmentation_amd"))
from segk import ops
from segk.ops import Act
dt = torch.bfloat16
K = 40
def bench(fa, fb, sa, sb):
    def run(which):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
        if which & 1:
            with torch.cuda.stream(sa):
                for _ in range(K): fa()
        if which & 2:
            with torch.cuda.stream(sb):
                for _ in range(K): fb()
        torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / K * 1e3
    for w in (1, 2, 3): run(w)
    return run(1), run(2), run(3)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for (H, Cin, Cout), (Hn, Cn) in [((96, 256, 256), (96, 256)), ((96, 256, 256), (192, 128)), ((48, 512, 512), (192, 128)), ((192, 128, 128), (192, 128)),
                                 ((96, 256, 256), (384, 64))]:
    N = 32
    x = Act(torch.randn(N, H, H, Cin, device="cuda").to(dt)); y = Act.empty(N, H, H, Cout, dt, "cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    wf, wb = ops.conv_pack(w, dt)
    z = Act(torch.randn(N, Hn, Hn, Cn, device="cuda").to(dt)); o = Act.empty(N, Hn, Hn, Cn, dt, "cuda")
    stats = torch.rand(2 * Cn, device="cuda") + 0.5; g = torch.ones(Cn, device="cuda"); b = torch.zeros(Cn, device="cuda")
    ta, tb, tab = bench(lambda: ops.conv_fwd(x, wf, Cout, y), lambda: ops.norm_apply(z, 1, stats, g, b, True, o), sa, sb)
    print(f"conv {Cin}->{Cout}@{H}^2 {ta:7.1f} us | norm_apply {Cn}ch@{Hn}^2 {tb:7.1f} us | together {tab:7.1f} us | sum {ta+tb:7.1f} | hidden {100*(ta+tb-tab)/min(ta,tb):5.1f} % of the shorter", flush=True)
