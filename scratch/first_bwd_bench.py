"""Times the two halves of segk_conv_first_norm_bwd separately, each twice in a row (cold / warm gradient)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-segmentation_amd"))
from segk import ops, _lib
from segk.ops import Act, _env, _p, workspace, query, call
N, H, W, C = 32, 384, 384, 64
x = torch.randn(N, 3, H, W, device="cuda"); w = torch.randn(C, 3, 3, 3, device="cuda") * 0.3; b = torch.randn(C, device="cuda") * 0.1
gamma = torch.rand(C, device="cuda") + 0.5; beta = torch.randn(C, device="cuda") * 0.3
stats = torch.empty(2 * C, device="cuda")
part = ops.conv_first_norm(1, x, w, b)
ops.norm_stats_from_partials(part, 1, N * H * W, 1e-5, stats)
g = Act(torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16))
junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")       # 1 GiB written between runs: evicts the memory-side cache
dw = torch.empty(C, 3, 3, 3, device="cuda"); dg = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda")
dev, st = _env(x)
ws = workspace(query("segk_conv_first_norm_bwd_workspace", N, H, W), x.device)
args = (x.data_ptr(), N, 3, H, W, w.data_ptr(), b.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), g.ptr, g.ld, dw.data_ptr(), dg.data_ptr(), db.data_ptr())
def run(flag):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call("segk_conv_first_norm_bwd", *args, flag, ws.data_ptr(), ws.numel(), dev, st); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
for name, flag in (("norm half (sums pass + reduction)", 4), ("convolution half (dz + weight gradient + reduction)", 8)):
    for rep in range(3):
        junk.fill_(rep); torch.cuda.synchronize()
        cold = run(flag); warm = run(flag); warm2 = run(flag)
        print(f"{name}: cold {cold:7.1f} us, again {warm:7.1f} us, again {warm2:7.1f} us", flush=True)
