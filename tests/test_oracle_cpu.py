"""CPU tests: the oracle restatement reproduces the golden vectors captured from the reference, and the
product's host-side surface (module tree, state_dict layout, CPU-tensor path, error behaviour) mirrors it."""
import json
import os

import numpy as np
import pytest
import torch

from util import VARIANTS, input_channels, report, step_loss, variant_oracle, variant_product

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_pin_recorded():
    pin = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_pin.json")))
    assert pin["all_bitwise_equal"] and all(pin["checks"].values())
    assert len(pin["checks"]) >= 40


def test_oracle_losses_match_golden(golden):
    from oracle import ref_losses as RL
    from oracle.cases import loss_cases, make_loss_inputs
    g = golden["losses"]
    for name, C, N, side, kw in loss_cases():
        if side > 64:
            continue
        z, t = make_loss_inputs(name, C, N, side, kw)
        dkw = {k: kw[k] for k in ("no_sft", "is_all_chann") if k in kw}
        fkw = {k: kw[k] for k in ("gamma", "reduction") if k in kw}
        if "alpha" in kw:
            fkw["alpha"] = torch.tensor(kw["alpha"])
        zin = torch.softmax(z, 1) if kw.get("no_sft") else z
        a = zin.clone().requires_grad_(True)
        ld = RL.dice_loss_sq(a, t, **dkw)
        ld.backward()
        assert np.array_equal(ld.detach().numpy(), g[f"{name}/dice"])
        assert np.array_equal(a.grad.numpy(), g[f"{name}/dice_grad"])
        b = z.clone().requires_grad_(True)
        lf = RL.focal_loss_with_logits(b, t, **fkw)
        lf.backward()
        assert np.array_equal(lf.detach().numpy(), g[f"{name}/focal"])
        assert np.array_equal(b.grad.numpy(), g[f"{name}/focal_grad"])
        assert np.array_equal(RL.dice_coefficient(z, t).numpy(), g[f"{name}/metric"])
        # closed forms used by the HIP backward agree with autograd (double precision)
        if not dkw and not fkw:
            cf = RL.dice_focal_closed_form_grad(z, t)
            report(f"{name}/closed_form", cf.float(), a.grad + b.grad, rtol=1e-4, atol=1e-7)


def test_aux_dice_losses_oracle_and_host_path():
    """Self-adjusting and log-cosh Dice: the oracle reproduces the reference's vectors bit for bit and the product's
    CPU-tensor path agrees with it."""
    from oracle import ref_losses as RL
    from oracle.cases import loss_cases, make_loss_inputs
    from loss.dice import dice_loss_sa, logcoshdiceloss
    g = np.load(os.path.join(ROOT, "tests", "golden", "losses_aux.npz"))
    seen = 0
    for name, C, N, side, kw in loss_cases():
        if f"{name}/sa" not in g.files:
            continue
        seen += 1
        z, t = make_loss_inputs(name, C, N, side, kw)
        nos = bool(kw.get("no_sft"))
        zin = torch.softmax(z, 1) if nos else z
        for tag, f_or, f_host in (("sa", RL.dice_loss_sa, dice_loss_sa), ("logcosh", RL.logcoshdiceloss, logcoshdiceloss)):
            a = zin.clone().requires_grad_(True)
            la = f_or(a, t, no_sft=nos)
            la.backward()
            assert np.array_equal(la.detach().numpy(), g[f"{name}/{tag}"]) and np.array_equal(a.grad.numpy(), g[f"{name}/{tag}_grad"])
            b = zin.clone().requires_grad_(True)
            lb = f_host(b, t, no_sft=nos)
            lb.backward()
            report(f"{name}/{tag} host", lb, la, rtol=1e-6, atol=1e-7)
            report(f"{name}/{tag} host grad", b.grad, a.grad, rtol=1e-5, atol=1e-9)
    assert seen >= 5


def test_edge_dice_loss_oracle_and_host_path():
    """Edge-aware Dice (loss/hausdorff.py:71-128): the oracle reproduces the reference's vectors bit for bit and the
    product's CPU-tensor path (one grouped convolution, |InstanceNorm|) agrees with it."""
    from oracle import ref_losses as RL
    from oracle.cases import loss_cases, make_loss_inputs
    from loss.edge_loss import get_hot_enc
    from loss.hausdorff import edge_conv_loss, get_edg_c
    g = np.load(os.path.join(ROOT, "tests", "golden", "losses_edge.npz"))
    seen = 0
    for name, C, N, side, kw in loss_cases():
        if f"{name}/edge_loss" not in g.files:
            continue
        seen += 1
        z, t = make_loss_inputs(name, C, N, side, kw)
        a = z.clone().requires_grad_(True)
        la = RL.edge_conv_loss(a, t)
        la.backward()
        assert np.array_equal(RL.get_edg_c(z).numpy(), g[f"{name}/edge_map"]) and np.array_equal(la.detach().numpy(), g[f"{name}/edge_loss"])
        assert np.array_equal(a.grad.numpy(), g[f"{name}/edge_grad"])
        assert torch.equal(get_hot_enc(t), RL.get_hot_enc(t))
        b = z.clone().requires_grad_(True)
        report(f"{name}/edge host map", get_edg_c(b), torch.from_numpy(g[f"{name}/edge_map"]), rtol=1e-5, atol=1e-5)
        lb = edge_conv_loss(b, t)
        lb.backward()
        report(f"{name}/edge host", lb, la, rtol=1e-5, atol=1e-6)
        report(f"{name}/edge host grad", b.grad, a.grad, rtol=1e-4, atol=1e-4 * float(a.grad.abs().max()))
    assert seen == 2
    assert get_edg_c(torch.zeros(2, 8, 8)).shape == (2, 1, 8, 8)          # label images become one plane


def test_remaining_loss_helpers_oracle_and_host_path():
    """dice_coefficientold (loss/dice.py:186-256) and the focal helpers (loss/focal.py:5-106): oracle bit-identical to the
    reference's vectors, product CPU path equal to the oracle."""
    from oracle import ref_losses as RL
    from oracle.cases import loss_cases, make_loss_inputs
    from loss.dice import dice_coefficientold
    from loss.focal import _focal_loss, _general_focal_loss, reduce
    g = np.load(os.path.join(ROOT, "tests", "golden", "losses_misc.npz"))
    seen = 0
    for name, C, N, side, kw in loss_cases():
        if f"{name}/dice_old" not in g.files:
            continue
        seen += 1
        z, t = make_loss_inputs(name, C, N, side, kw)
        assert np.array_equal(RL.dice_coefficientold(z, t).numpy(), g[f"{name}/dice_old"])
        assert np.array_equal(RL.dice_coefficientold(z, t, is_list_bat=True).numpy(), g[f"{name}/dice_old_list"])
        report(f"{name}/dice_old host", dice_coefficientold(z, t), torch.from_numpy(g[f"{name}/dice_old"]), rtol=1e-6, atol=1e-7)
        report(f"{name}/dice_old list host", dice_coefficientold(z, t, is_list_bat=True), torch.from_numpy(g[f"{name}/dice_old_list"]), rtol=1e-6, atol=1e-7)
        report(f"{name}/dice_old mask host", dice_coefficientold(z.argmax(1, keepdim=True), t, channelcnt=C),
               torch.from_numpy(g[f"{name}/dice_old_from_mask"]), rtol=1e-6, atol=1e-7)
        p, ce = torch.softmax(z, 1), torch.nn.functional.cross_entropy(z, t, reduction="none")
        assert np.array_equal(RL.focal_from_probabilities(p, t, 2.0, ce, "elementwise_mean").numpy(), g[f"{name}/focal_from_probs"])
        report(f"{name}/focal host", _focal_loss(p=p, t=t, gamma=2.0, loss_val=ce, reduction="elementwise_mean"),
               torch.from_numpy(g[f"{name}/focal_from_probs"]), rtol=1e-6, atol=1e-8)
        hot = torch.zeros_like(p).scatter_(1, t.unsqueeze(1), 1)
        assert _general_focal_loss(p, hot, 2.0, ce, reduction="none").shape == p.shape
    assert seen == 4
    assert float(reduce(torch.ones(3), "sum")) == 3.0
    with pytest.raises(ValueError, match="Reduction parameter unknown"):
        reduce(torch.ones(3), "bogus")


def test_oracle_known_answers(golden):
    from oracle import ref_losses as RL
    z = torch.zeros(2, 4, 8, 8)
    t = torch.zeros(2, 8, 8, dtype=torch.long)
    t[:, :4, :] = 1
    t[0, 4:, :4] = 2
    assert abs(float(RL.focal_loss_with_logits(z, t)) - 0.75 / 4 * np.log(4)) < 1e-7
    assert abs(float(RL.dice_loss_sq(z, t)) - 0.67541826) < 1e-7
    assert float(RL.dice_loss_sq(z, torch.zeros_like(t))) == 1.0
    assert float(RL.dice_coefficient(z, torch.zeros_like(t))) == 0.0
    with pytest.raises(RuntimeError):
        RL.dice_loss_sq(torch.zeros(1, 4, 8, 12), torch.zeros(1, 8, 12, dtype=torch.long))
    with pytest.raises(ValueError, match="Reduction parameter unknown"):
        RL.focal_loss_with_logits(z, t, reduction="bogus")
    with pytest.raises(TypeError):
        RL.dice_coefficient(z, torch.zeros(2, 4, 8, 8))


def _product(name):
    if name in VARIANTS:
        return variant_product(name)
    from models.unetvgg import UNetVgg, UNetVggInOut
    from models.unetplusplus import Nested_UNet
    return {
        "unetvgg_c4": lambda: UNetVgg(out_classes=4),
        "unetvgg_c3": lambda: UNetVgg(out_classes=3),
        "unetvgg_c4_convtrans": lambda: UNetVgg(out_classes=4, if_convtrans=True),
        "unetvgginout_c4": lambda: UNetVggInOut(in_classes=1, out_classes=4),
        "nestedunet_c4": lambda: Nested_UNet(in_ch=1, out_ch=4),
        "detectmid": lambda: __import__("models.resnt", fromlist=["DetectMid"]).DetectMid(in_classes=1),
        "detectmid_regr": lambda: __import__("models.resnt", fromlist=["DetectMid"]).DetectMid(in_classes=1, if_regr=True, out_classes=1),
    }[name]()


def _oracle(name):
    if name in VARIANTS:
        return variant_oracle(name)
    from oracle import ref_models as RM
    return {
        "unetvgg_c4": lambda: RM.RefUNetVgg(out_classes=4),
        "unetvgg_c3": lambda: RM.RefUNetVgg(out_classes=3),
        "unetvgg_c4_convtrans": lambda: RM.RefUNetVgg(out_classes=4, if_convtrans=True),
        "unetvgginout_c4": lambda: RM.RefUNetVgg(out_classes=4, in_classes=1),
        "nestedunet_c4": lambda: RM.RefNestedUNet(in_ch=1, out_ch=4),
        "detectmid": lambda: RM.RefDetectMid(in_classes=1),
        "detectmid_regr": lambda: RM.RefDetectMid(in_classes=1, if_regr=True, out_classes=1),
    }[name]()


ALL = ["unetvgg_c4", "unetvgg_c3", "unetvgg_c4_convtrans", "unetvgginout_c4", "nestedunet_c4", "detectmid", "detectmid_regr"] + list(VARIANTS)


@pytest.mark.parametrize("name", ALL)
def test_state_dict_layout_matches_reference(name):
    keys = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")))[name]
    sd = _product(name).state_dict()
    assert list(sd.keys()) == list(keys.keys())
    for k, v in sd.items():
        assert list(v.shape) == keys[k], k
    # a reference-layout checkpoint loads strictly
    _product(name).load_state_dict(_oracle(name).state_dict(), strict=True)


@pytest.mark.parametrize("name", ["unetvgg_c4", "nestedunet_c4", "unetsimple_cs32_c3", "unetnaive_convtrans_pool_c3", "unetlimmulti_c4",
                                  "unetsimple_multi_cs32_c3", "spidernet_multi_cs32_c3", "unettetra_cs32_c4"])
def test_oracle_and_host_path_match_golden(name, golden):
    """One train step on CPU: the oracle equals the golden vectors bit for bit, and the product's CPU-tensor
    path (plain torch.nn modules holding the same parameters) equals the oracle."""
    from oracle import ref_losses as RL
    from oracle.ref_models import hash_init_, synthetic_batch
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    g = golden["models"]
    in_ch = input_channels(name)
    x, t = synthetic_batch(2, 64, 3 if name.endswith("c3") else 4, in_ch=in_ch, tag=f"model:{name}")
    torch.set_num_threads(8)
    m = hash_init_(_oracle(name)).train()
    logits, feats, loss = step_loss(m(x), t, RL.dice_loss_sq, RL.focal_loss_with_logits)
    loss.backward()
    for i, f in enumerate(feats):
        assert np.array_equal(f.detach()[..., ::4, ::4].numpy(), g[f"{name}/feat{i}_sub"])
    assert np.array_equal(logits.detach().numpy(), g[f"{name}/logits"])
    assert np.array_equal(loss.detach().numpy(), g[f"{name}/loss"])
    assert np.array_equal(logits.argmax(1).to(torch.uint8).numpy(), g[f"{name}/mask"])
    p = hash_init_(_product(name)).train()
    lp, featsp, lossp = step_loss(p(x), t, dice_loss_sq, focal_loss_with_logits)
    assert len(featsp) == len(feats)
    lossp.backward()
    report("host logits", lp, logits, rtol=1e-5, atol=1e-5)
    report("host loss", lossp, loss, rtol=1e-5, atol=1e-6)
    for (k, a), (_, b) in zip(p.named_parameters(), m.named_parameters()):
        if b.grad is None:          # never used by forward (SpiderNet.conv5)
            assert a.grad is None, k
            continue
        report(k, a.grad, b.grad, rtol=1e-3, atol=1e-5)


def test_detectmid_host_path_matches_oracle():
    """3-D classifier on CPU tensors (plain torch.nn modules holding the parameters): one small train step equals the
    oracle's (which make_golden.py pinned bit for bit to the reference's models/resnt.py + models/densenet3d.py)."""
    import torch.nn.functional as F
    from oracle.ref_models import RefDetectMid, hash_init_, synthetic_volumes
    from models.resnt import DetectMid
    x, t = synthetic_volumes(2, 16, 64, tag="model:host")
    torch.set_num_threads(8)
    outs = []
    for mk in (RefDetectMid, DetectMid):
        m = hash_init_(mk(in_classes=1)).train()
        y = m(x)
        F.cross_entropy(y, t).backward()
        outs.append((y.detach(), {k: p.grad for k, p in m.named_parameters()}, dict(m.named_buffers())))
    assert torch.equal(outs[0][0], outs[1][0])
    for k, g in outs[0][1].items():
        assert (g is None and outs[1][1][k] is None) or torch.equal(g, outs[1][1][k]), k
    for k, b in outs[0][2].items():
        assert torch.equal(b, outs[1][2][k]), k


def test_resizing_wrappers_layout_and_host_path(tmp_path):
    """UNetVgg256 / UNetVggTwo / UNetVggWholeImg (models/unetvgg.py:107-155, 216-265): state_dict layout of the reference,
    and the CPU-tensor path equal to the oracle's on a small image."""
    import models.unetvgg as M
    from oracle import ref_models as RM
    keys = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")))
    for name, mk, mk_or in (("unetvgg256", M.UNetVgg256, RM.RefUNetVgg256), ("unetvggtwo", M.UNetVggTwo, RM.RefUNetVggTwo)):
        sd = mk().state_dict()
        assert list(sd.keys()) == list(keys[name].keys()) and all(list(v.shape) == keys[name][k] for k, v in sd.items())
    torch.set_num_threads(8)
    x, _ = RM.synthetic_batch(1, 64, 3, in_ch=3, tag="model:wrap_host")
    a, b = RM.hash_init_(M.UNetVgg256()).eval(), RM.hash_init_(RM.RefUNetVgg256()).eval()
    with torch.no_grad():
        assert torch.equal(a(x[..., :40]), b(x[..., :40]))
    ck = tmp_path / "ck.pt"
    torch.save({"state_dict": a.unetvgg.state_dict()}, ck)
    w = M.UNetVggWholeImg(str(ck)).eval()
    with torch.no_grad():
        assert torch.equal(w(x), a.unetvgg(x))


def test_constructor_surface():
    from models.unetvgg import UNetVgg, UNetVggInOut, make_decoder_block
    from models.unetplusplus import Nested_UNet, conv_block_nested
    m = UNetVgg()
    assert m.final.out_channels == 3 and m.patch_train is False
    assert UNetVgg(out_classes=4, patch_train=True).patch_train is True
    assert UNetVggInOut(in_classes=2).conv0.in_channels == 2
    assert Nested_UNet().final.out_channels == 3
    blk = make_decoder_block(8, 16, 4)
    assert [type(c).__name__ for c in blk] == ["Conv2d", "ReLU", "Conv2d", "InstanceNorm2d", "Upsample", "ReLU"]
    assert [type(c).__name__ for c in make_decoder_block(8, 16, 4, if_convtrans=True)] == ["Conv2d", "ReLU", "ConvTranspose2d", "ReLU"]
    assert [type(c).__name__ for c in make_decoder_block(8, 16, 4, noupsam=True)] == ["Conv2d", "ReLU", "Conv2d", "InstanceNorm2d", "ReLU"]
    assert blk(torch.zeros(1, 8, 4, 4)).shape == (1, 4, 8, 8)
    assert conv_block_nested(3, 8, 8)(torch.zeros(2, 3, 4, 4)).shape == (2, 8, 4, 4)
    from models.unetvgg import UNetNaive, UNetSimple, UNetWithoutCenter, make_center, make_encoder_block
    assert [type(c).__name__ for c in make_center(512)] == ["MaxPool2d", "Conv2d", "ReLU", "Conv2d", "Upsample", "ReLU"]
    assert [type(c).__name__ for c in make_center(512, max_pool=False, iftrans2d=True)] == ["Conv2d", "Conv2d", "ReLU", "ConvTranspose2d", "ReLU"]
    assert UNetNaive(in_classes=2).conv1[0].in_channels == 2 and UNetWithoutCenter().conv5[4].out_channels == 256
    enc = make_encoder_block(8, 16, 16, middle_channels2=16)
    assert [type(c).__name__ for c in enc] == ["Conv2d"] + ["Conv2d", "InstanceNorm2d", "ReLU"] * 3 and enc[0].stride == (2, 2)
    assert len(make_encoder_block(8, 16, 16, convs3=False, max_pool=True)) == 7
    assert enc(torch.zeros(1, 8, 8, 8)).shape == (1, 16, 4, 4)
    from models.unetvgg import UNetLimMultiOut, UNetNaiveMultiOut
    assert torch.is_tensor(UNetLimMultiOut(multiout=True)(torch.zeros(1, 1, 16, 16))[1])
    assert [f.shape[1] for f in UNetNaiveMultiOut()(torch.zeros(1, 1, 16, 16))[1]] == [256, 256]
    from models.unetvgg import UNetSimpleReduced, UNetSkip
    assert UNetSkip().channelscale == 32 and torch.is_tensor(UNetSkip(multiout=True)(torch.zeros(1, 1, 16, 16))[1])
    with pytest.raises(NameError):
        UNetSimpleReduced(channelscale=16, multiout=True)(torch.zeros(1, 1, 16, 16))
    u = UNetSimple(channelscale=16, multiout=True)
    out, feats = u(torch.zeros(1, 1, 16, 16))
    assert out.shape == (1, 3, 16, 16) and [f.shape[1] for f in feats] == [8, 8, 16, 64]
    with pytest.raises(RuntimeError, match="pretrained"):
        UNetVgg(pretrained=True)
    from models.densenet3d import generate_model
    from models.resnt import DetectMid
    d = DetectMid(in_classes=2, out_classes=3)
    assert d.first.in_channels == 2 and d.last.out_features == 80 and d.lastregr.out_features == 3 and d.patch_train is False
    assert len(d.rsnet.features.denseblock3) == 32 and d.rsnet.classifier.in_features == 1664
    assert generate_model(121).classifier.in_features == 1024
    with pytest.raises(AssertionError):
        generate_model(50)


def test_gpu_tensors_never_fall_back(monkeypatch):
    """Without the HIP library a CUDA forward must raise rather than compute elsewhere."""
    from segk import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsegk.so")
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.SegkError, match="missing"):
        _lib.lib()


def test_nestedunet_deep_supervision_host_path_equals_oracle():
    """Four-head UNet++ (an extension the reference does not contain; oracle = this repository's torch restatement, not
    pinned by the reference): the product's CPU path and the oracle agree bit for bit, and the default network keeps
    the reference's single `final` head."""
    import torch
    from models.unetplusplus import Nested_UNet, deep_supervision_loss
    from oracle import ref_losses as RL
    from oracle import ref_models as RM
    x, t = RM.synthetic_batch(1, 32, 4, in_ch=1, tag="deepsup-cpu")
    a = RM.hash_init_(Nested_UNet(in_ch=1, out_ch=4, deep_supervision=True)).train()
    b = RM.hash_init_(RM.RefNestedUNet(in_ch=1, out_ch=4, deep_supervision=True)).train()
    assert [k for k in a.state_dict() if k.startswith("final")] == [f"final{j}.{w}" for j in range(1, 5) for w in ("weight", "bias")]
    oa, ob = a(x), b(x)
    assert len(oa) == 4 and all(torch.equal(p, q) for p, q in zip(oa, ob))
    la = deep_supervision_loss(oa, t, [RL.dice_loss_sq, RL.focal_loss_with_logits])
    lb = deep_supervision_loss(ob, t, [RL.dice_loss_sq, RL.focal_loss_with_logits])
    assert torch.equal(la, lb)
    assert "final.weight" in Nested_UNet(in_ch=1, out_ch=4).state_dict() and "final1.weight" not in Nested_UNet().state_dict()


def test_oracle_reproduces_the_well_conditioned_fixture():
    """tests/golden/wellcond.npz comes from the reference import (oracle/make_golden_wellcond.py); the oracle must give the
    same logits, loss and gradient summaries bit for bit on every case (UNetVgg with either decoder, Nested_UNet), and the pin
    file must say so."""
    import json
    import os
    import numpy as np
    import torch
    from oracle import ref_losses as RL
    from oracle import wellcond as WC
    torch.set_num_threads(8)               # the fixture was made with 8 threads (the reduction order of the CPU kernels)
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(root, "wellcond.npz"))
    pins = json.load(open(os.path.join(root, "oracle_pin.json")))["checks"]
    assert set(WC.TAGS) <= set(WC.CASES) and len(WC.CASES) == 4
    for tag in WC.CASES:
        assert pins[f"model/{tag}"] is True, tag
        x, t = WC.batch(tag)
        m = WC.oracle_model(tag)
        lg = m(x)
        loss = RL.dice_loss_sq(lg, t) + RL.focal_loss_with_logits(lg, t)
        loss.backward()
        assert np.array_equal(lg.detach().numpy(), g[f"{tag}/logits"]) and float(loss) == float(g[f"{tag}/loss"]), tag
        for k, p in m.named_parameters():
            flat = p.grad.flatten()
            stride = max(1, flat.numel() // 4096)
            assert np.array_equal(flat[::stride][:4096].numpy(), g[f"{tag}/grad/{k}/sub"]), (tag, k)
        assert float(g[f"{tag}/min_relu_margin"]) > 5e-7, tag
        # the reference's own fp32 gradients against exact arithmetic: 5e-6..8e-6 for UNetVgg, 9e-5 for UNet++ (recorded, not hidden)
        assert float(g[f"{tag}/reference_fp32_vs_fp64_worst_grad_rel_l2"]) < (1e-4 if "unetpp" in tag else 2e-5), tag
