"""Micro-benchmark of streaming kernels at UNet-VGG shapes (batch 32, bf16): python scratch/stream_bench.py [names]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-segmentation_amd"))
from segk import ops
from segk.ops import Act
N = 32
dt = torch.bfloat16

def timeit(f, reps=int(os.environ.get('REPS', 5))):
    for _ in range(max(2, reps // 2)): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def act(H, C, ld=None, c0=0):
    ld = ld or C
    return Act(torch.randn(N, H, H, ld, device="cuda").to(dt), c0, C)

which = sys.argv[1:] or ["up_bwd", "up_fwd", "pool_bwd", "pool_fwd"]
# decoder up-sampling layers: (low-res H, C, ld of the concat buffer that holds the high-res output)
UPS = [(192, 32, 96), (96, 64, 192), (48, 256, 512), (24, 256, 768), (12, 256, 768)]
for H, C, ld in UPS:
    for strided in (True, False):
        l = ld if strided else C
        x = act(H, C); u = act(2 * H, C, l); du = act(2 * H, C, l); dx = act(H, C)
        st = torch.randn(2 * N * C, device="cuda").abs() + 0.5
        mb_b = (N * 4 * H * H * C * 2 + N * H * H * C * 2) / 1e6
        mb_f = (N * 4 * H * H * C * 2 + N * H * H * C * 2) / 1e6
        line = f"up {H:3d}->{2*H:3d} C={C:3d} ld={l:3d} |"
        if "up_bwd" in which:
            t = timeit(lambda: ops.upsample_bwd(du, None, dx, relu=False, align=False)); line += f" bwd {t:7.1f} us {mb_b/t:5.2f} TB/s |"
        if "up_fwd" in which:
            t = timeit(lambda: ops.upsample_fwd(x, u, stats=st, relu=True, align=False)); line += f" fwd {t:7.1f} us {mb_f/t:5.2f} TB/s |"
        print(line, flush=True)
POOLS = [(384, 64, 160), (192, 128, 192), (96, 256, 320), (48, 512, 768), (24, 512, 768)]
for H, C, ld in POOLS:
    x = act(H, C, ld, ld - C); dxx = act(H, C, ld, ld - C); p = act(H // 2, C); dp = act(H // 2, C)
    line = f"pool {H:3d} C={C:3d} ld={ld:3d} |"
    sz = N * H * H * C * 2 / 1e6
    if "pool_fwd" in which:
        t = timeit(lambda: ops.maxpool_fwd(x, p)); line += f" fwd {t:7.1f} us {1.25*sz/t:5.2f} TB/s |"
    if "pool_bwd" in which:
        t = timeit(lambda: ops.maxpool_bwd(x, dp, dxx, accumulate=True)); line += f" bwd(acc) {t:7.1f} us {3.25*sz/t:5.2f} TB/s |"
    print(line, flush=True)
if "norm" in which:
    for H, C in [(384, 64), (192, 128), (96, 256), (48, 512)]:
        x = act(H, C); dy = act(H, C); out = act(H, C); dx = act(H, C)
        stats = torch.empty(2 * C, device="cuda"); g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
        dg = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda")
        sz = N * H * H * C * 2 / 1e6
        ops.norm_stats(x, 1, 1e-5, stats)
        t1 = timeit(lambda: ops.norm_stats(x, 1, 1e-5, stats))
        t2 = timeit(lambda: ops.norm_apply(x, 1, stats, g, b, True, out))
        t3 = timeit(lambda: ops.norm_bwd(dy, x, 1, stats, g, b, True, dx, dgamma=dg, dbeta=db))
        print(f"norm {H:3d} C={C:3d} | stats {t1:6.1f} us {sz/t1:5.2f} TB/s | apply {t2:6.1f} us {2*sz/t2:5.2f} TB/s | bwd {t3:6.1f} us {5*sz/t3:5.2f} TB/s |", flush=True)
