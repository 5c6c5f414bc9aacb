"""bf16 HIP path on the well-conditioned cases: logits / loss / gradient rel-L2 against the fp32 oracle, next to the oracle run
under torch's bf16 autocast (the same rounding points on the CPU)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-segmentation_amd"))
from oracle import ref_losses as RL, wellcond as WC
from loss.dice import dice_loss_sq
from loss.focal import focal_loss_with_logits
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for tag in WC.CASES:
    x, t = WC.batch(tag)
    ref = WC.oracle_model(tag)
    lr = ref(x); lossr = RL.dice_loss_sq(lr, t) + RL.focal_loss_with_logits(lr, t); lossr.backward()
    auto = WC.oracle_model(tag)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        la = auto(x)
    lossa = RL.dice_loss_sq(la.float(), t) + RL.focal_loss_with_logits(la.float(), t); lossa.backward()
    net = WC.hip_model(tag).cuda().set_compute_dtype(torch.bfloat16)
    lg = net(x.cuda()); loss = dice_loss_sq(lg, t.cuda()) + focal_loss_with_logits(lg, t.cuda()); loss.backward()
    torch.cuda.synchronize()
    eh, ea = [], []
    for (k, p), (_, q), (_, qa) in zip(net.named_parameters(), ref.named_parameters(), auto.named_parameters()):
        if float(q.grad.norm()) < 1e-6: continue
        eh.append((rel(p.grad.cpu(), q.grad), k)); ea.append((rel(qa.grad, q.grad), k))
    eh.sort(reverse=True); ea.sort(reverse=True)
    print(f"{tag}: logits hip {rel(lg.cpu(), lr):.3e} autocast {rel(la.float(), lr):.3e} | loss hip {abs(float(loss)-float(lossr))/abs(float(lossr)):.3e} "
          f"autocast {abs(float(lossa)-float(lossr))/abs(float(lossr)):.3e} | grads hip worst {eh[0][0]:.3e} ({eh[0][1]}) median {eh[len(eh)//2][0]:.3e} "
          f"| autocast worst {ea[0][0]:.3e} ({ea[0][1]}) median {ea[len(ea)//2][0]:.3e} | hip tensors over 2e-2: {sum(1 for e,_ in eh if e > 2e-2)}/{len(eh)}", flush=True)
