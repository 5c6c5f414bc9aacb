import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-segmentation_amd"))
from oracle import ref_models as RM, ref_losses as RL, wellcond as WC
from models.unetvgg import UNetVgg
from loss.dice import dice_loss_sq
from loss.focal import focal_loss_with_logits
for i in [3, 8, 6, 10, 7, 5, 11]:
    tag = f"wellcond:{i}"
    x, t = WC.batch(tag)
    refs = {}
    for dt in (torch.float32, torch.float64):
        m = WC.condition_(RM.hash_init_(RM.RefUNetVgg(out_classes=4))).to(dt).train()
        lg = m(x.to(dt)); (RL.dice_loss_sq(lg, t) + RL.focal_loss_with_logits(lg, t)).backward()
        refs[dt] = (m, lg.detach())
    net = WC.condition_(RM.hash_init_(UNetVgg(out_classes=4))).cuda().train().set_compute_dtype(torch.float32)
    lg = net(x.cuda()); (dice_loss_sq(lg, t.cuda()) + focal_loss_with_logits(lg, t.cuda())).backward()
    torch.cuda.synchronize()
    w32 = w64 = r = 0
    for (k, p), (_, q32), (_, q64) in zip(net.named_parameters(), refs[torch.float32][0].named_parameters(), refs[torch.float64][0].named_parameters()):
        s = float(q64.grad.norm())
        if s < 1e-9: continue
        w32 = max(w32, float((p.grad.cpu().double() - q32.grad.double()).norm()) / s)
        w64 = max(w64, float((p.grad.cpu().double() - q64.grad).norm()) / s)
        r = max(r, float((q32.grad.double() - q64.grad).norm()) / s)
    print(tag, "logits rel", float((lg.detach().cpu() - refs[torch.float32][1]).norm() / refs[torch.float32][1].norm()),
          "worst grad rel-L2: hip vs ref32 %.2e  hip vs fp64 %.2e  ref32 vs fp64 %.2e" % (w32, w64, r), flush=True)
