"""Host-side enqueue time of one training step vs its GPU time (is the step launch-bound?)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
from bench import synthetic
from loss.dice import dice_loss_sq
from loss.focal import focal_loss_with_logits
from models.unetvgg import UNetVgg
from segk.flat import FlatModel
from segk.optim import Adam
from segk import _lib
B = int(os.environ.get("B", 32))
model = UNetVgg(out_classes=4).cuda().train().set_compute_dtype(torch.bfloat16)
flat = FlatModel(model); opt = Adam(flat, lr=1e-4)
x, t = synthetic(B, 384, 4, "cuda")
def step():
    logits = model(x)
    loss = dice_loss_sq(logits, t) + focal_loss_with_logits(logits, t)
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
n0 = _lib.launches
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"batch {B}: host enqueue {1e3*(t1-t0)/10:.2f} ms/step, with GPU drain {1e3*(t2-t0)/10:.2f} ms/step, C-ABI calls/step {(_lib.launches-n0)/10:.0f}")
