"""Does running the main chain on a high-priority stream (weight gradients on the normal-priority side stream) shorten the step?
    python scratch/prio_step.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
from bench import synthetic
from loss.dice import dice_loss_sq
from loss.focal import focal_loss_with_logits
from models.unetvgg import UNetVgg
from segk.flat import FlatModel
from segk.optim import Adam
from segk import engine
B = 32
model = UNetVgg(out_classes=4).cuda().train().set_compute_dtype(torch.bfloat16)
flat = FlatModel(model); opt = Adam(flat, lr=1e-4)
x, t = synthetic(B, 384, 4, "cuda")
def step():
    logits = model(x)
    loss = dice_loss_sq(logits, t) + focal_loss_with_logits(logits, t)
    loss.backward(); opt.step()
def timed(n=30):
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
for rnd in range(2):
    engine._WGRAD_STREAM = True
    print(f"default stream + side stream: {timed():.3f} ms/step", flush=True)
    hi = torch.cuda.Stream(priority=-1)
    hi.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(hi):
        print(f"high-priority main + normal side: {timed():.3f} ms/step", flush=True)
    torch.cuda.synchronize()
    engine._WGRAD_STREAM = False
    print(f"one stream: {timed():.3f} ms/step", flush=True)
