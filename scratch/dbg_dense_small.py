"""Small DenseNet (few rows per BatchNorm channel): HIP fp32 gradients vs torch fp32/fp64 on the CPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import torch
import torch.nn.functional as F
from oracle.ref_models import RefDenseNet3D, hash_init_, hash_uniform
from models.densenet3d import DenseNet

cfg = dict(n_input_channels=3, conv1_t_size=3, block_config=tuple(int(v) for v in sys.argv[1].split(",")), num_init_features=int(sys.argv[2]), num_classes=10)
shape = tuple(int(v) for v in sys.argv[3].split(","))
x = hash_uniform("dbg:x", (2, 3) + shape, -1.7, 1.7)
t = torch.tensor([3, 7])
m = hash_init_(DenseNet(**cfg)).cuda().train().set_compute_dtype(torch.float32)
out = m(x.cuda()); F.cross_entropy(out, t.cuda()).backward(); torch.cuda.synchronize()
res = {}
for dt in (torch.float32, torch.float64):
    o = hash_init_(RefDenseNet3D(**cfg)).to(dt).train()
    y = o(x.to(dt)); F.cross_entropy(y, t).backward()
    res[dt] = (y.detach(), {k: p.grad for k, p in o.named_parameters() if p.grad is not None})
def rl(a, b): return float((a.detach().double().cpu() - b.double()).norm() / (b.double().norm() + 1e-30))
print("out: hip vs f64", rl(out, res[torch.float64][0]), " or32 vs f64", rl(res[torch.float32][0], res[torch.float64][0]))
for k, p in m.named_parameters():
    g64 = res[torch.float64][1][k]
    print("hip %.3e  ref32 %.3e  %s %s" % (rl(p.grad, g64), rl(res[torch.float32][1][k], g64), k, tuple(p.shape)))
