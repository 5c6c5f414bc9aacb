import sys, torch
sys.path.insert(0, "image-segmentation_amd")
from segk import ops
x = torch.randn(32, 3, 384, 384, device="cuda")
for _ in range(3):
    ops.im2col3x3(x, torch.bfloat16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.im2col3x3(x, torch.bfloat16)
e1.record(); torch.cuda.synchronize()
print("im2col3x3 32x3x384x384 bf16: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
