"""tests/golden/wellcond.npz: one training step of the REFERENCE's UNetVgg (bilinear and transposed-convolution decoder) and Nested_UNet on the well-conditioned cases of
oracle/wellcond.py, with every parameter gradient summarised so that the whole tensor is pinned in a few KB:
its norm, 4096 strided samples and four +-1 hash projections.  Also pins the oracle restatement bit for bit on
these cases (tests/golden/oracle_pin.json).  Build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_wellcond.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "stubs"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

from oracle import ref_losses as RL  # noqa: E402
from oracle import ref_models as RM  # noqa: E402
from oracle import wellcond as WC  # noqa: E402

torch.set_num_threads(8)


def summary(name, g):
    """norm, strided samples and hash projections of a gradient tensor (shared with the tests)."""
    flat = g.detach().double().flatten().cpu()
    stride = max(1, flat.numel() // 4096)
    signs = [RM.hash_uniform(f"proj{i}:{name}", (flat.numel(),), -1, 1).sign().double() for i in range(4)]
    return {"norm": np.float64(flat.norm().item()), "sub": flat[::stride][:4096].float().numpy(),
            "proj": np.array([float(flat @ s) for s in signs], dtype=np.float64)}


def main():
    import importlib
    uv = importlib.import_module("models.unetvgg")
    upp = importlib.import_module("models.unetplusplus")
    dice = importlib.import_module("loss.dice")
    focal = importlib.import_module("loss.focal")
    assert uv.__file__.startswith("/root/reference") and upp.__file__.startswith("/root/reference")
    out, pin = {}, {}
    for tag, (kind, kw, _) in WC.CASES.items():
        x, t = WC.batch(tag)
        ref_net = uv.UNetVgg(out_classes=4, **kw) if kind == "unetvgg" else upp.Nested_UNet(in_ch=1, out_ch=4)
        m_ref = WC.condition_(RM.hash_init_(ref_net)).train()
        m_or = WC.oracle_model(tag)
        la = m_ref(x)
        lossa = dice.dice_loss_sq(la, t) + focal.focal_loss_with_logits(la, t)
        lossa.backward()
        lb = m_or(x)
        lossb = RL.dice_loss_sq(lb, t) + RL.focal_loss_with_logits(lb, t)
        lossb.backward()
        same = torch.equal(la, lb) and torch.equal(lossa, lossb)
        for (k, p), (_, q) in zip(m_ref.named_parameters(), m_or.named_parameters()):
            same = same and torch.equal(p.grad, q.grad)
        pin[f"model/{tag}"] = bool(same)
        m64 = WC.oracle_model(tag).double()
        l64 = m64(x.double())
        (RL.dice_loss_sq(l64, t) + RL.focal_loss_with_logits(l64, t)).backward()
        worst = max(float((p.grad.double() - q.grad).norm() / q.grad.norm()) for (_, p), (_, q) in
                    zip(m_ref.named_parameters(), m64.named_parameters()) if float(q.grad.norm()) > 1e-9)
        margin, near = WC.relu_margin(m64.eval().train(), x.double())
        out[f"{tag}/logits"] = la.detach().numpy()
        out[f"{tag}/loss"] = lossa.detach().numpy()
        out[f"{tag}/min_relu_margin"] = np.float64(margin)          # smallest |pre-ReLU| / mean |pre-ReLU| of its tensor
        out[f"{tag}/values_within_1e-5_of_threshold"] = np.int64(near)
        out[f"{tag}/reference_fp32_vs_fp64_worst_grad_rel_l2"] = np.float64(worst)
        for k, p in m_ref.named_parameters():
            for part, v in summary(k, p.grad).items():
                out[f"{tag}/grad/{k}/{part}"] = v
        print(tag, "pinned" if same else "NOT PINNED", "margin %.2e" % margin, "near", near, "reference fp32 vs fp64 %.2e" % worst, flush=True)
    np.savez_compressed(os.path.join(GOLD, "wellcond.npz"), **out)
    path = os.path.join(GOLD, "oracle_pin.json")
    with open(path) as f:
        doc = json.load(f)
    target = doc["checks"] if isinstance(doc.get("checks"), dict) else doc
    target.update(pin)
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    assert all(pin.values()), pin


if __name__ == "__main__":
    main()
