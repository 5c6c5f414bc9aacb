"""Is a training step launch-bound on the host, and does replaying it as a HIP graph help?
    MODEL=detectmid|unetvgg B=<batch> python scratch/graph_step.py
Prints host enqueue time, eager step time, and -- if the step can be captured -- the replayed graph's step time."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
from bench import synthetic
from segk.flat import FlatModel
from segk.optim import Adam
from segk import _lib

name = os.environ.get("MODEL", "detectmid")
dt = torch.bfloat16
if name == "detectmid":
    from models.resnt import DetectMid
    B = int(os.environ.get("B", 12))
    model = DetectMid(in_classes=1)
    g = torch.Generator().manual_seed(20)
    x = torch.randn(B, 1, 64, 256, 256, generator=g).cuda()
    t = torch.randint(0, 64, (B,), generator=g).cuda()
    lossf = lambda out: torch.nn.functional.cross_entropy(out, t)
else:
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from models.unetvgg import UNetVgg
    B = int(os.environ.get("B", 32))
    model = UNetVgg(out_classes=4)
    x, t = synthetic(B, 384, 4, "cuda")
    lossf = lambda out: dice_loss_sq(out, t) + focal_loss_with_logits(out, t)
model = model.cuda().train().set_compute_dtype(dt)
flat = FlatModel(model); opt = Adam(flat, lr=1e-4)

def step():
    loss = lossf(model(x))
    loss.backward(); opt.step()
    return loss

for _ in range(3): step()
torch.cuda.synchronize()
n0 = _lib.launches
t0 = time.perf_counter()
for _ in range(10): loss = step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{name} batch {B}: host enqueue {1e3*(t1-t0)/10:.2f} ms/step, eager step {1e3*(t2-t0)/10:.2f} ms, C-ABI calls/step {(_lib.launches-n0)/10:.0f}, "
      f"loss {float(loss):.5f}", flush=True)

if os.environ.get("CAPTURE", "1") == "1":
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gl = step()
    torch.cuda.synchronize()
    print("captured", flush=True)
    for _ in range(3): graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): graph.replay()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"graph replay {1e3*(t1-t0)/10:.2f} ms/step, loss {float(gl):.5f}", flush=True)
