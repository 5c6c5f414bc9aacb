"""Generate tests/golden/*.npz from the REFERENCE ITSELF and pin the oracle restatement to it.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

It puts oracle/stubs (stand-ins for the absent torchvision / rising / pytorch_lightning / scikit-image packages)
and /root/reference on sys.path, imports the reference's models/unetvgg.py, models/unetplusplus.py,
loss/dice.py and loss/focal.py, runs them on hash-initialised weights and synthetic batches, and

 1. asserts that oracle/ref_models.py and oracle/ref_losses.py reproduce every output BIT FOR BIT
    (same ATen calls in the same order), writing the verdict to tests/golden/oracle_pin.json;
 2. stores the reference outputs as small fixtures that the CPU and GPU tests compare against.

Nothing of the reference's source travels: fixtures are inputs' recipes (tags for the hash
initialiser) and output arrays only.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "stubs"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from oracle import ref_losses as RL  # noqa: E402
from oracle import ref_models as RM  # noqa: E402
from oracle.cases import loss_cases, make_loss_inputs  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(20)


def import_reference():
    import importlib
    mods = {}
    for name in ("models.unetvgg", "models.unetplusplus", "models.resnt", "models.densenet3d", "loss.dice", "loss.focal"):
        mods[name] = importlib.import_module(name)
        assert mods[name].__file__.startswith(REF), mods[name].__file__
    return mods


def bits_equal(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b)


# ------------------------------------------------------------------------------------ losses
def golden_losses(ref, pin):
    out = {}
    dice_ref, focal_ref = ref["loss.dice"], ref["loss.focal"]
    for name, C, N, side, kw in loss_cases():
        z, t = make_loss_inputs(name, C, N, side, kw)
        dkw = {k: kw[k] for k in ("no_sft", "is_all_chann") if k in kw}
        fkw = {k: kw[k] for k in ("gamma", "reduction") if k in kw}
        if "alpha" in kw:
            fkw["alpha"] = torch.tensor(kw["alpha"])
        res = {}
        for tag, fn_ref, fn_or, k in (("dice", dice_ref.dice_loss_sq, RL.dice_loss_sq, dkw),
                                      ("focal", focal_ref.focal_loss_with_logits, RL.focal_loss_with_logits, fkw)):
            zin = torch.softmax(z, 1) if (tag == "dice" and kw.get("no_sft")) else z
            a = zin.clone().requires_grad_(True)
            la = fn_ref(a, t, **k)
            la.sum().backward()
            b = zin.clone().requires_grad_(True)
            lb = fn_or(b, t, **k)
            lb.sum().backward()
            pin[f"loss/{name}/{tag}"] = bool(bits_equal(la.detach(), lb.detach()) and bits_equal(a.grad, b.grad))
            res[tag] = la.detach().numpy()
            g = a.grad
            res[tag + "_grad"] = g.numpy() if side <= 64 else g[..., ::16, ::16].contiguous().numpy()
            res[tag + "_grad_sum"] = np.float64(g.double().sum().item())
            res[tag + "_grad_abs"] = np.float64(g.double().abs().sum().item())
        # metric (non-differentiable)
        m_ref = dice_ref.dice_coefficient(z, t)
        m_or = RL.dice_coefficient(z, t)
        pin[f"loss/{name}/metric"] = bool(bits_equal(m_ref, m_or))
        res["metric"] = m_ref.numpy()
        res["metric_list"] = dice_ref.dice_coefficient(z, t, is_list_bat=True).numpy()
        for k2, v in res.items():
            out[f"{name}/{k2}"] = v
    # 4-D one-hot target path (dice.py:53-54) and the closed-form known answers
    z = torch.zeros(2, 4, 8, 8)
    t = torch.zeros(2, 8, 8, dtype=torch.long)
    t[:, :4, :] = 1
    t[0, 4:, :4] = 2
    out["ka/zero_logits_dice"] = dice_ref.dice_loss_sq(z, t).numpy()
    out["ka/zero_logits_focal"] = focal_ref.focal_loss_with_logits(z, t).numpy()
    t2 = t.clone()
    t2[1] = 0
    out["ka/zero_logits_dice_img1bg"] = dice_ref.dice_loss_sq(z, t2).numpy()
    out["ka/allbg_dice"] = dice_ref.dice_loss_sq(z, torch.zeros_like(t)).numpy()
    out["ka/allbg_metric"] = dice_ref.dice_coefficient(z, torch.zeros_like(t)).numpy()
    hot = torch.zeros(2, 4, 8, 8).scatter(1, t.unsqueeze(1), 1)
    zz = RM.hash_uniform("loss:onehot4d", (2, 4, 8, 8), -2, 2)
    out["ka/onehot4d_dice"] = dice_ref.dice_loss_sq(zz, hot).numpy()
    pin["loss/onehot4d"] = bool(bits_equal(dice_ref.dice_loss_sq(zz, hot), RL.dice_loss_sq(zz, hot)))
    np.savez_compressed(os.path.join(GOLD, "losses.npz"), **out)


def golden_aux_losses(ref, pin):
    """Self-adjusting and log-cosh Dice (loss/dice.py:275-349) on the loss cases that use the default Dice options."""
    dice_ref = ref["loss.dice"]
    out = {}
    for name, C, N, side, kw in loss_cases():
        if side > 64 or any(k in kw for k in ("is_all_chann", "gamma", "alpha", "reduction")):
            continue
        z, t = make_loss_inputs(name, C, N, side, kw)
        nos = bool(kw.get("no_sft"))
        zin = torch.softmax(z, 1) if nos else z
        for tag, fn_ref, fn_or in (("sa", dice_ref.dice_loss_sa, RL.dice_loss_sa), ("logcosh", dice_ref.logcoshdiceloss, RL.logcoshdiceloss)):
            a = zin.clone().requires_grad_(True)
            la = fn_ref(a, t, no_sft=nos)
            la.backward()
            b = zin.clone().requires_grad_(True)
            lb = fn_or(b, t, no_sft=nos)
            lb.backward()
            pin[f"loss/{name}/{tag}"] = bool(bits_equal(la.detach(), lb.detach()) and bits_equal(a.grad, b.grad))
            out[f"{name}/{tag}"] = la.detach().numpy()
            out[f"{name}/{tag}_grad"] = a.grad.numpy()
    np.savez_compressed(os.path.join(GOLD, "losses_aux.npz"), **out)


def golden_edge_losses(pin):
    """Edge-aware Dice (loss/hausdorff.py:71-128) on the three-class loss cases: edge map of the logits, loss, dlogits."""
    import importlib
    hd = importlib.import_module("loss.hausdorff")
    assert hd.__file__.startswith(REF), hd.__file__
    out = {}
    for name, C, N, side, kw in loss_cases():
        if C != 3:
            continue
        z, t = make_loss_inputs(name, C, N, side, kw)
        a = z.clone().requires_grad_(True)
        ea = hd.get_edg_c(a)
        la = hd.edge_conv_loss(a, t)
        la.backward()
        b = z.clone().requires_grad_(True)
        eb = RL.get_edg_c(b)
        lb = RL.edge_conv_loss(b, t)
        lb.backward()
        pin[f"loss/{name}/edge"] = bool(bits_equal(ea.detach(), eb.detach()) and bits_equal(la.detach(), lb.detach()) and bits_equal(a.grad, b.grad)
                                        and bits_equal(hd.get_edg_c(hd.get_hot_enc(t)), RL.get_edg_c(RL.get_hot_enc(t))))
        out[f"{name}/edge_map"] = ea.detach().numpy()
        out[f"{name}/edge_target"] = hd.get_edg_c(hd.get_hot_enc(t)).numpy()
        out[f"{name}/edge_loss"] = la.detach().numpy()
        out[f"{name}/edge_grad"] = a.grad.numpy()
    np.savez_compressed(os.path.join(GOLD, "losses_edge.npz"), **out)


def golden_misc_losses(ref, pin):
    """The remaining public helpers of loss/dice.py and loss/focal.py: dice_coefficientold (:186-256) and
    _focal_loss (:77-106) on softmax probabilities with a cross-entropy map."""
    dice_ref, focal_ref = ref["loss.dice"], ref["loss.focal"]
    out = {}
    for name, C, N, side, kw in loss_cases():
        if side > 64 or kw:
            continue
        z, t = make_loss_inputs(name, C, N, side, kw)
        a, b = dice_ref.dice_coefficientold(z, t), RL.dice_coefficientold(z, t)
        la, lb = dice_ref.dice_coefficientold(z, t, is_list_bat=True), RL.dice_coefficientold(z, t, is_list_bat=True)
        ma = dice_ref.dice_coefficientold(z.argmax(1, keepdim=True), t, channelcnt=C)
        mb = RL.dice_coefficientold(z.argmax(1, keepdim=True), t, channelcnt=C)
        p = torch.softmax(z, 1)
        ce = torch.nn.functional.cross_entropy(z, t, reduction="none")
        fa = focal_ref._focal_loss(p=p, t=t, gamma=2.0, loss_val=ce, reduction="elementwise_mean")
        fb = RL.focal_from_probabilities(p, t, 2.0, ce, "elementwise_mean")
        pin[f"loss/{name}/misc"] = bool(bits_equal(a, b) and bits_equal(la, lb) and bits_equal(ma, mb) and bits_equal(fa, fb))
        out[f"{name}/dice_old"], out[f"{name}/dice_old_list"] = a.numpy(), la.numpy()
        out[f"{name}/dice_old_from_mask"], out[f"{name}/focal_from_probs"] = ma.numpy(), fa.numpy()
    np.savez_compressed(os.path.join(GOLD, "losses_misc.npz"), **out)


# ------------------------------------------------------------------------------------ models
def model_cases(ref):
    uv, up = ref["models.unetvgg"], ref["models.unetplusplus"]
    return [
        ("unetvgg_c4", lambda: uv.UNetVgg(out_classes=4), lambda: RM.RefUNetVgg(out_classes=4), 3),
        ("unetvgg_c3", lambda: uv.UNetVgg(out_classes=3), lambda: RM.RefUNetVgg(out_classes=3), 3),
        ("unetvgg_c4_convtrans", lambda: uv.UNetVgg(out_classes=4, if_convtrans=True),
         lambda: RM.RefUNetVgg(out_classes=4, if_convtrans=True), 3),
        ("unetvgginout_c4", lambda: uv.UNetVggInOut(in_classes=1, out_classes=4),
         lambda: RM.RefUNetVgg(out_classes=4, in_classes=1), 1),
        ("nestedunet_c4", lambda: up.Nested_UNet(in_ch=1, out_ch=4), lambda: RM.RefNestedUNet(in_ch=1, out_ch=4), 1),
    ]


FEATURE_WEIGHT = 0.05


def split_outputs(out):
    """model output -> (logits, list of extra feature maps): the multi-output U-Nets return (logits, tensor or list)."""
    if not isinstance(out, tuple):
        return out, []
    return out[0], list(out[1]) if isinstance(out[1], (list, tuple)) else [out[1]]


def run_step(model, x, t, losses):
    """Loss of the step: the given losses on the logits, plus FEATURE_WEIGHT * mean(f^2) for every extra feature map a
    multi-output network returns (so that gradients flow back through those outputs as well)."""
    model.train()
    model.zero_grad()
    logits, feats = split_outputs(model(x))
    loss = sum(fn(logits, t) for fn in losses)
    for f in feats:
        loss = loss + FEATURE_WEIGHT * (f * f).mean()
    loss.backward()
    run_step.feats = [f.detach() for f in feats]
    return logits.detach(), loss.detach()


def _model_fixture(name, mk_ref, mk_or, in_ch, n_cls, out, keys, pin, dice_ref, focal_ref):
    """One training step of the reference network `mk_ref()` and of its oracle restatement `mk_or()` on the same
    hash-initialised weights and synthetic batch: bitwise comparison into `pin`, reference outputs into `out`."""
    x, t = RM.synthetic_batch(2, 64, n_cls, in_ch=in_ch, tag=f"model:{name}")
    m_ref = RM.hash_init_(mk_ref())
    m_or = RM.hash_init_(mk_or())
    assert list(m_ref.state_dict().keys()) == list(m_or.state_dict().keys()), name
    for (ka, va), (kb, vb) in zip(m_ref.state_dict().items(), m_or.state_dict().items()):
        assert va.shape == vb.shape and torch.equal(va, vb), (name, ka)
    keys[name] = {k: list(v.shape) for k, v in m_ref.state_dict().items()}
    la, lossa = run_step(m_ref, x, t, [dice_ref.dice_loss_sq, focal_ref.focal_loss_with_logits])
    feats_a = run_step.feats
    lb, lossb = run_step(m_or, x, t, [RL.dice_loss_sq, RL.focal_loss_with_logits])
    same = bits_equal(la, lb) and bits_equal(lossa, lossb) and len(feats_a) == len(run_step.feats)
    for i, (fa, fb) in enumerate(zip(feats_a, run_step.feats)):
        same = same and bits_equal(fa, fb)
        out[f"{name}/feat{i}_sub"] = fa[..., ::4, ::4].contiguous().numpy()
    for (ka, pa), (_, pb) in zip(m_ref.named_parameters(), m_or.named_parameters()):
        same = same and ((pa.grad is None and pb.grad is None) or bits_equal(pa.grad, pb.grad))    # None: never used by forward
    for (ka, ba), (_, bb) in zip(m_ref.named_buffers(), m_or.named_buffers()):
        same = same and bits_equal(ba, bb)
    pin[f"model/{name}"] = bool(same)
    out[f"{name}/logits"] = la.numpy()
    out[f"{name}/loss"] = lossa.numpy()
    out[f"{name}/mask"] = la.argmax(1).to(torch.uint8).numpy()
    top2 = la.topk(2, dim=1).values
    out[f"{name}/near_tie_1e-3"] = np.int64(((top2[:, 0] - top2[:, 1]) < 1e-3).sum().item())
    for k, p in m_ref.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.flatten()
        out[f"{name}/grad_norm/{k}"] = np.float64(g.double().norm().item())
        out[f"{name}/grad_head/{k}"] = g[:16].numpy().copy()
    for k, b in m_ref.named_buffers():
        if not k.endswith("num_batches_tracked"):
            out[f"{name}/buf/{k}"] = b.numpy().copy()
    # eval-mode logits after the step (running statistics path)
    m_ref.eval()
    with torch.no_grad():
        le, _ = split_outputs(m_ref(x))
    out[f"{name}/eval_logits_sub"] = le[..., ::4, ::4].contiguous().numpy()
    out[f"{name}/eval_mask"] = le.argmax(1).to(torch.uint8).numpy()


def golden_models(ref, pin):
    dice_ref, focal_ref = ref["loss.dice"], ref["loss.focal"]
    out, keys = {}, {}
    for name, mk_ref, mk_or, in_ch in model_cases(ref):
        n_cls = 3 if name.endswith("c3") else 4
        _model_fixture(name, mk_ref, mk_or, in_ch, n_cls, out, keys, pin, dice_ref, focal_ref)
    # one full-size slice: UNetVgg 384x384, N=1, C=4
    uv = ref["models.unetvgg"]
    x, t = RM.synthetic_batch(1, 384, 4, in_ch=3, tag="model:unetvgg384")
    m_ref = RM.hash_init_(uv.UNetVgg(out_classes=4))
    m_or = RM.hash_init_(RM.RefUNetVgg(out_classes=4))
    la, lossa = run_step(m_ref, x, t, [dice_ref.dice_loss_sq, focal_ref.focal_loss_with_logits])
    lb, lossb = run_step(m_or, x, t, [RL.dice_loss_sq, RL.focal_loss_with_logits])
    pin["model/unetvgg384"] = bool(bits_equal(la, lb) and bits_equal(lossa, lossb))
    out["unetvgg384/mask_packed"] = np.packbits(la.argmax(1).to(torch.uint8).numpy().astype(np.uint8), bitorder="little")  # 2 bits would do; keep simple
    out["unetvgg384/mask"] = la.argmax(1).to(torch.uint8).numpy()
    out["unetvgg384/logits_sub"] = la[..., ::8, ::8].contiguous().numpy()
    out["unetvgg384/loss"] = lossa.numpy()
    top2 = la.topk(2, dim=1).values
    out["unetvgg384/near_tie_1e-3"] = np.int64(((top2[:, 0] - top2[:, 1]) < 1e-3).sum().item())
    for k, p in m_ref.named_parameters():
        out[f"unetvgg384/grad_norm/{k}"] = np.float64(p.grad.double().norm().item())
    del out["unetvgg384/mask_packed"]
    np.savez_compressed(os.path.join(GOLD, "models.npz"), **out)
    with open(os.path.join(GOLD, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)


# ------------------------------------------------------------------------------------ further U-Net variants
def variant_cases(ref):
    uv = ref["models.unetvgg"]
    return [
        ("unetsimple_c4", lambda: uv.UNetSimple(out_classes=4, in_classes=1), lambda: RM.RefUNetSimple(out_classes=4, in_classes=1), 1),
        ("unetsimple_cs32_c3", lambda: uv.UNetSimple(out_classes=3, in_classes=1, channelscale=32),
         lambda: RM.RefUNetSimple(out_classes=3, in_classes=1, channelscale=32), 1),
        ("unetsimple_flat_cs16_c4", lambda: uv.UNetSimple(out_classes=4, in_classes=1, channelscale=16, noupdownsam=True),
         lambda: RM.RefUNetSimple(out_classes=4, in_classes=1, channelscale=16, noupdownsam=True), 1),
        ("unetnaive_c4", lambda: uv.UNetNaive(out_classes=4, in_classes=1), lambda: RM.RefUNetNaive(out_classes=4, in_classes=1), 1),
        ("unetnaive_convtrans_pool_c3", lambda: uv.UNetNaive(out_classes=3, in_classes=1, if_convtrans=True, ifmaxpool=True),
         lambda: RM.RefUNetNaive(out_classes=3, in_classes=1, if_convtrans=True, ifmaxpool=True), 1),
        ("unetnocenter_c4", lambda: uv.UNetWithoutCenter(out_classes=4, in_classes=1),
         lambda: RM.RefUNetNaive(out_classes=4, in_classes=1, center=False), 1),
        ("unetsimple_multi_cs32_c3", lambda: uv.UNetSimple(out_classes=3, in_classes=1, channelscale=32, multiout=True),
         lambda: RM.RefUNetSimple(out_classes=3, in_classes=1, channelscale=32, multiout=True), 1),
        ("unetnaivemulti_c4", lambda: uv.UNetNaiveMultiOut(out_classes=4, in_classes=1), lambda: RM.RefUNetFourLevel(out_classes=4), 1),
        ("unetlimmulti_c4", lambda: uv.UNetLimMultiOut(out_classes=4, in_classes=1, multiout=True),
         lambda: RM.RefUNetFourLevel(out_classes=4, channel_depth=16), 1),
        ("spidernet_c4", lambda: uv.SpiderNet(out_classes=4, in_classes=1), lambda: RM.RefSpiderNet(out_classes=4), 1),
        ("spidernet_multi_cs32_c3", lambda: uv.SpiderNet(out_classes=3, in_classes=1, channelscale=32, multiout=True),
         lambda: RM.RefSpiderNet(out_classes=3, channelscale=32, multiout=True), 1),
        ("unettetra_cs32_c4", lambda: uv.UNetTetra(out_classes=4, in_classes=1, channelscale=32),
         lambda: RM.RefUNetTetra(out_classes=4, channelscale=32), 1),
        ("unetskip_multi_c4", lambda: uv.UNetSkip(out_classes=4, in_classes=1, multiout=True),
         lambda: RM.RefUNetSimple(out_classes=4, in_classes=1, channelscale=32, multiout="dec4"), 1),
        ("unetsimplereduced_cs32_c3", lambda: uv.UNetSimpleReduced(out_classes=3, in_classes=1, channelscale=32),
         lambda: RM.RefUNetSimpleReduced(out_classes=3, in_classes=1, channelscale=32), 1),
    ]


def golden_variants(ref, pin):
    """UNetSimple (models/unetvgg.py:651-728): default width, half width, and the variant without re-scaling;
    UNetNaive (:373-440) with both bottleneck flavours; UNetWithoutCenter (:443-506)."""
    import warnings
    warnings.filterwarnings("ignore", message=".*num_features.*")
    dice_ref, focal_ref = ref["loss.dice"], ref["loss.focal"]
    out, keys = {}, {}
    for name, mk_ref, mk_or, in_ch in variant_cases(ref):
        _model_fixture(name, mk_ref, mk_or, in_ch, 3 if name.endswith("c3") else 4, out, keys, pin, dice_ref, focal_ref)
    np.savez_compressed(os.path.join(GOLD, "variants.npz"), **out)
    kf = os.path.join(GOLD, "state_dict_keys.json")
    allkeys = json.load(open(kf)) if os.path.exists(kf) else {}
    allkeys.update(keys)
    with open(kf, "w") as f:
        json.dump(allkeys, f, indent=0)


# ------------------------------------------------------------------------------------ re-sizing wrappers
def golden_wrappers(ref, pin):
    """UNetVgg256 (models/unetvgg.py:107-132) and UNetVggTwo (:216-265, with the label-driven crop): one training step each."""
    uv, dice_ref, focal_ref = ref["models.unetvgg"], ref["loss.dice"], ref["loss.focal"]
    out, keys = {}, {}
    cases = [("unetvgg256", uv.UNetVgg256, RM.RefUNetVgg256, 3, False), ("unetvggtwo", uv.UNetVggTwo, RM.RefUNetVggTwo, 1, True)]
    for name, mk_ref, mk_or, in_ch, with_labels in cases:
        x, t = RM.synthetic_batch(2, 96, 3, in_ch=in_ch, tag=f"model:{name}")
        res = []
        for mk, dice, focal in ((mk_ref, dice_ref.dice_loss_sq, focal_ref.focal_loss_with_logits), (mk_or, RL.dice_loss_sq, RL.focal_loss_with_logits)):
            m = RM.hash_init_(mk()).train()
            y = m(x, t) if with_labels else m(x)
            loss = dice(y, t) + focal(y, t)
            loss.backward()
            res.append((m, y.detach(), loss.detach()))
        (ma, ya, la), (mb, yb, lb) = res
        assert list(ma.state_dict().keys()) == list(mb.state_dict().keys()), name
        same = bits_equal(ya, yb) and bits_equal(la, lb)
        for (_, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
            same = same and bits_equal(pa.grad, pb.grad)
        pin[f"model/{name}"] = bool(same)
        keys[name] = {k: list(v.shape) for k, v in ma.state_dict().items()}
        out[f"{name}/logits"] = ya.numpy()
        out[f"{name}/loss"] = la.numpy()
        for k, p_ in ma.named_parameters():
            out[f"{name}/grad_norm/{k}"] = np.float64(p_.grad.double().norm().item())
    np.savez_compressed(os.path.join(GOLD, "wrappers.npz"), **out)
    kf = os.path.join(GOLD, "state_dict_keys.json")
    allkeys = json.load(open(kf)) if os.path.exists(kf) else {}
    allkeys.update(keys)
    with open(kf, "w") as f:
        json.dump(allkeys, f, indent=0)


# ------------------------------------------------------------------------------------ 3-D classifier
# (2, 32, 128) volumes leave 2x2x2 voxels x 2 volumes = 16 samples per channel for the last block's BatchNorms; with
# fewer (a 16x64x64 volume ends at ONE voxel, 2 samples) the network is chaotic: the reference's own fp32 and fp64 runs
# then disagree by 30 % on the output, so such a case pins nothing.
VOLUME_CASES = [("detectmid", dict(), (2, 32, 128)), ("detectmid_regr", dict(if_regr=True, out_classes=1), (2, 32, 128))]


def golden_volumes(ref, pin):
    """DetectMid (models/resnt.py:8-45) on synthetic volumes with the loss of train_midlslice.py:29 (cross entropy on
    the 80 slice classes); the regression variant with an MSE loss on the squeezed output."""
    resnt = ref["models.resnt"]
    out, keys = {}, {}
    for name, kw, (n, depth, side) in VOLUME_CASES:
        x, t = RM.synthetic_volumes(n, depth, side, tag=f"model:{name}")
        m_ref = RM.hash_init_(resnt.DetectMid(in_classes=1, **kw))
        m_or = RM.hash_init_(RM.RefDetectMid(in_classes=1, **kw))
        assert list(m_ref.state_dict().keys()) == list(m_or.state_dict().keys()), name
        keys[name] = {k: list(v.shape) for k, v in m_ref.state_dict().items()}
        res = []
        for m in (m_ref, m_or):
            m.train()
            m.zero_grad()
            y = m(x)
            loss = torch.nn.functional.mse_loss(y, t.float() / 80) if kw.get("if_regr") else torch.nn.functional.cross_entropy(y, t)
            loss.backward()
            res.append((y.detach(), loss.detach()))
        same = bits_equal(res[0][0], res[1][0]) and bits_equal(res[0][1], res[1][1])
        for (ka, pa), (_, pb) in zip(m_ref.named_parameters(), m_or.named_parameters()):
            same = same and ((pa.grad is None and pb.grad is None) or bits_equal(pa.grad, pb.grad))
        for (ka, ba), (_, bb) in zip(m_ref.named_buffers(), m_or.named_buffers()):
            same = same and bits_equal(ba, bb)
        pin[f"model/{name}"] = bool(same)
        out[f"{name}/out"] = res[0][0].numpy()
        out[f"{name}/loss"] = res[0][1].numpy()
        for k, p_ in m_ref.named_parameters():
            if p_.grad is None:
                continue
            g = p_.grad.flatten()
            out[f"{name}/grad_norm/{k}"] = np.float64(g.double().norm().item())
            out[f"{name}/grad_head/{k}"] = g[:16].numpy().copy()
        for k, b in m_ref.named_buffers():
            if k.endswith("running_mean") or k.endswith("running_var"):
                out[f"{name}/bufsum/{k}"] = np.float64(b.double().sum().item())
        m_ref.eval()
        with torch.no_grad():
            out[f"{name}/eval_out"] = m_ref(x).numpy()
    np.savez_compressed(os.path.join(GOLD, "volumes.npz"), **out)
    kf = os.path.join(GOLD, "state_dict_keys.json")
    allkeys = json.load(open(kf)) if os.path.exists(kf) else {}
    allkeys.update(keys)
    with open(kf, "w") as f:
        json.dump(allkeys, f, indent=0)


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref = import_reference()
    pin = {}
    partial = [a for a in ("--volumes-only", "--aux-losses-only", "--variants-only", "--edge-losses-only", "--wrappers-only", "--misc-losses-only") if a in sys.argv]
    if partial:                           # refresh one fixture file without regenerating the others
        pf = os.path.join(GOLD, "oracle_pin.json")
        pin = json.load(open(pf))["checks"] if os.path.exists(pf) else {}
        if "--volumes-only" in partial:
            golden_volumes(ref, pin)
        if "--aux-losses-only" in partial:
            golden_aux_losses(ref, pin)
        if "--variants-only" in partial:
            golden_variants(ref, pin)
        if "--edge-losses-only" in partial:
            golden_edge_losses(pin)
        if "--wrappers-only" in partial:
            golden_wrappers(ref, pin)
        if "--misc-losses-only" in partial:
            golden_misc_losses(ref, pin)
    else:
        # a full run rewrites every check of its own; checks that other generator scripts recorded in the same file
        # (oracle/make_golden_wellcond.py: "model/wellcond*", oracle/run_reference_callers.py) are carried over
        pf = os.path.join(GOLD, "oracle_pin.json")
        if os.path.exists(pf):
            pin = {k: v for k, v in json.load(open(pf)).get("checks", {}).items() if k.startswith(("model/wellcond", "callers/"))}
        golden_losses(ref, pin)
        golden_aux_losses(ref, pin)
        golden_edge_losses(pin)
        golden_misc_losses(ref, pin)
        golden_models(ref, pin)
        golden_variants(ref, pin)
        golden_wrappers(ref, pin)
        golden_volumes(ref, pin)
    verdict = {"torch": torch.__version__, "numpy": np.__version__, "all_bitwise_equal": all(pin.values()), "checks": pin}
    with open(os.path.join(GOLD, "oracle_pin.json"), "w") as f:
        json.dump(verdict, f, indent=1, sort_keys=True)
    print(json.dumps(verdict, indent=1, sort_keys=True))
    if not verdict["all_bitwise_equal"]:
        raise SystemExit("oracle restatement differs from the reference")


if __name__ == "__main__":
    main()
