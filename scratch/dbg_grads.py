import sys, os, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'image-segmentation_amd')); sys.path.insert(0,os.path.join(ROOT,'tests'))
from test_gpu_models import run_step, oracle_step
for name in sys.argv[1:]:
    model, logits, loss = run_step(name, torch.float32)
    m32, l32, _ = oracle_step(name, torch.float32)
    m64, l64, _ = oracle_step(name, torch.float64)
    print(name, "logits err hip vs 64: %.2e  ref32 vs 64: %.2e" % (float((logits.cpu().double()-l64).abs().max()), float((l32.double()-l64).abs().max())))
    for (k,p),(_,q32),(_,q64) in zip(model.named_parameters(), m32.named_parameters(), m64.named_parameters()):
        s=float(q64.grad.norm())
        if s<1e-6: continue
        print("  %-20s hip %.2e ref32 %.2e" % (k, float((p.grad.double().cpu()-q64.grad).norm())/s, float((q32.grad.double()-q64.grad).norm())/s))
