"""HIP fp32 gradients of DetectMid vs the CPU oracle in fp32 and fp64 (noise floor of the reference itself)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import torch
import torch.nn.functional as F
from oracle.ref_models import RefDetectMid, hash_init_, synthetic_volumes
from models.resnt import DetectMid

shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 32, 128)
x, t = synthetic_volumes(*shape, tag="model:detectmid")
torch.set_num_threads(16)
m = hash_init_(DetectMid()).cuda().train().set_compute_dtype(torch.float32)
out = m(x.cuda()); F.cross_entropy(out, t.cuda()).backward(); torch.cuda.synchronize()
res = {}
for dt in (torch.float32, torch.float64):
    t0 = time.time()
    o = hash_init_(RefDetectMid()).to(dt).train()
    y = o(x.to(dt)); F.cross_entropy(y, t).backward()
    res[dt] = (y.detach(), {k: p.grad for k, p in o.named_parameters() if p.grad is not None})
    print(dt, "oracle step", round(time.time() - t0, 1), "s", flush=True)
def rl(a, b): return float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-30))
print("out: hip vs f64", rl(out, res[torch.float64][0]), " or32 vs f64", rl(res[torch.float32][0], res[torch.float64][0]))
rows = []
for k, p in m.named_parameters():
    if k not in res[torch.float64][1]: continue
    g64 = res[torch.float64][1][k]
    rows.append((rl(p.grad, g64), rl(res[torch.float32][1][k], g64), k))
rows.sort(reverse=True)
for r in rows[:12]: print("hip %.3e  ref32 %.3e  %s" % r)
import statistics
print("median hip %.3e ref32 %.3e" % (statistics.median(r[0] for r in rows), statistics.median(r[1] for r in rows)))
rows.sort(key=lambda r: -(r[0] / (r[1] + 1e-9)))
print("--- by ratio")
for r in rows[:25]: print("hip %.3e  ref32 %.3e  %s" % r)
