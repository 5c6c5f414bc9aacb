"""GPU parity of the fused softmax + Dice + focal kernels against the golden vectors captured from the
reference (tests/golden/losses.npz) and against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

from util import report

pytestmark = pytest.mark.gpu


def _cases():
    from oracle.cases import loss_cases
    return loss_cases()


def _inputs(name, C, N, side, kw):
    from oracle.cases import make_loss_inputs
    return make_loss_inputs(name, C, N, side, kw)


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c[0])
def test_losses_vs_golden(case, golden):
    from loss.dice import dice_coefficient, dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from segk import _lib
    name, C, N, side, kw = case
    g = golden["losses"]
    z, t = _inputs(name, C, N, side, kw)
    before = _lib.launches
    dkw = {k: kw[k] for k in ("no_sft", "is_all_chann") if k in kw}
    fkw = {k: kw[k] for k in ("gamma", "reduction") if k in kw}
    if "alpha" in kw:
        fkw["alpha"] = torch.tensor(kw["alpha"])
    zin = torch.softmax(z, 1) if kw.get("no_sft") else z
    a = zin.clone().cuda().requires_grad_(True)
    ld = dice_loss_sq(a, t.cuda(), **dkw)
    ld.backward()
    report(f"{name}/dice", ld.detach().cpu(), torch.from_numpy(g[f"{name}/dice"]), rtol=1e-5, atol=1e-6)
    want = torch.from_numpy(g[f"{name}/dice_grad"])
    got = a.grad.cpu() if side <= 64 else a.grad.cpu()[..., ::16, ::16]
    report(f"{name}/dice_grad", got, want, rtol=1e-4, atol=1e-4 * float(want.abs().max()) + 1e-12)
    assert abs(float(a.grad.double().sum()) - float(g[f"{name}/dice_grad_sum"])) <= 1e-4 * float(g[f"{name}/dice_grad_abs"]) + 1e-9
    b = z.clone().cuda().requires_grad_(True)
    lf = focal_loss_with_logits(b, t.cuda(), **fkw)
    lf.backward()
    report(f"{name}/focal", lf.detach().cpu(), torch.from_numpy(g[f"{name}/focal"]), rtol=1e-5, atol=1e-6)
    want = torch.from_numpy(g[f"{name}/focal_grad"])
    got = b.grad.cpu() if side <= 64 else b.grad.cpu()[..., ::16, ::16]
    report(f"{name}/focal_grad", got, want, rtol=1e-4, atol=1e-4 * float(want.abs().max()) + 1e-12)
    assert abs(float(b.grad.double().sum()) - float(g[f"{name}/focal_grad_sum"])) <= 1e-4 * float(g[f"{name}/focal_grad_abs"]) + 1e-9
    m = dice_coefficient(z.cuda(), t.cuda())
    report(f"{name}/metric", m.cpu(), torch.from_numpy(g[f"{name}/metric"]), rtol=1e-6, atol=1e-7)
    ml = dice_coefficient(z.cuda(), t.cuda(), is_list_bat=True)
    report(f"{name}/metric_list", ml.cpu(), torch.from_numpy(g[f"{name}/metric_list"]), rtol=1e-6, atol=1e-7)
    assert _lib.launches > before, "the HIP library was not called"


def test_known_answers_and_errors(golden):
    from loss.dice import dice_coefficient, dice_loss_sq
    from loss.focal import focal_loss_with_logits
    g = golden["losses"]
    z = torch.zeros(2, 4, 8, 8, device="cuda")
    t = torch.zeros(2, 8, 8, dtype=torch.long)
    t[:, :4, :] = 1
    t[0, 4:, :4] = 2
    assert abs(float(focal_loss_with_logits(z, t.cuda())) - 0.75 / 4 * np.log(4)) < 1e-6
    assert abs(float(dice_loss_sq(z, t.cuda())) - float(g["ka/zero_logits_dice"])) < 1e-6
    t2 = t.clone(); t2[1] = 0
    assert abs(float(dice_loss_sq(z, t2.cuda())) - float(g["ka/zero_logits_dice_img1bg"])) < 1e-6
    assert float(dice_loss_sq(z, torch.zeros_like(t).cuda())) == 1.0
    assert float(dice_coefficient(z, torch.zeros_like(t).cuda())) == 0.0
    # explicit one-hot (4-D) target path
    from oracle.ref_models import hash_uniform
    hot = torch.zeros(2, 4, 8, 8).scatter(1, t.unsqueeze(1), 1)
    zz = hash_uniform("loss:onehot4d", (2, 4, 8, 8), -2, 2)
    assert abs(float(dice_loss_sq(zz.cuda(), hot.cuda())) - float(g["ka/onehot4d_dice"])) < 1e-6
    # error behaviour of the reference surface
    with pytest.raises(RuntimeError):
        dice_loss_sq(torch.zeros(1, 4, 8, 12, device="cuda"), torch.zeros(1, 8, 12, dtype=torch.long, device="cuda"))
    with pytest.raises(ValueError, match="Reduction parameter unknown"):
        focal_loss_with_logits(z, t.cuda(), reduction="bogus")
    with pytest.raises(RuntimeError):
        dice_loss_sq(z, t.cuda().int())


def test_loss_bf16_logits_and_closed_form():
    """bf16 logits (dlogits comes back in bf16) and the double-precision closed-form gradient."""
    from oracle.ref_losses import dice_focal_closed_form_grad
    from oracle.ref_models import hash_uniform, synthetic_batch
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    z = hash_uniform("loss:bf16", (3, 4, 32, 32), -3, 3).to(torch.bfloat16)
    _, t = synthetic_batch(3, 32, 4, tag="loss:bf16")
    a = z.clone().cuda().requires_grad_(True)
    loss = dice_loss_sq(a, t.cuda()) + focal_loss_with_logits(a, t.cuda()).float()
    loss.backward()
    assert a.grad.dtype == torch.bfloat16
    want = dice_focal_closed_form_grad(z.float(), t)
    report("bf16_grad", a.grad.float().cpu(), want.float(), rtol=2e-2, atol=2e-2 * float(want.abs().max()))


def test_loss_full_size_properties():
    """BASELINE-size batch (32 x 4 x 384 x 384): properties that need no oracle run at this size."""
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    torch.manual_seed(20)
    N, C, S = 32, 4, 384
    z = torch.randn(N, C, S, S, device="cuda", requires_grad=True)
    t = torch.randint(0, C, (N, S, S), device="cuda")
    t[::8] = 0
    lf = focal_loss_with_logits(z, t)
    ld = dice_loss_sq(z, t)
    (lf + ld).backward()
    g = z.grad
    # softmax-composed gradients sum to zero over the class axis at every pixel
    assert float(g.sum(1).abs().max()) < 1e-6
    # all-background images receive no Dice gradient: their total gradient is the focal one alone
    z2 = z.detach().clone().requires_grad_(True)
    focal_loss_with_logits(z2, t).backward()
    assert torch.allclose(g[::8], z2.grad[::8], rtol=1e-5, atol=1e-9)
    # focal with gamma = 0 degenerates to the mean cross entropy (the C per-class weights are all 1)
    lf0 = focal_loss_with_logits(z.detach(), t, gamma=0.0)
    ce = torch.nn.functional.cross_entropy(z.detach(), t)
    assert abs(float(lf0) - float(ce)) < 1e-5 * float(ce)
    # determinism: same inputs, same bits
    assert float(dice_loss_sq(z.detach(), t)) == float(ld)


def _aux_cases():
    from oracle.cases import loss_cases
    return [c for c in loss_cases() if c[3] <= 64 and not any(k in c[4] for k in ("is_all_chann", "gamma", "alpha", "reduction"))]


@pytest.mark.parametrize("case", _aux_cases(), ids=lambda c: c[0])
def test_self_adjusting_and_logcosh_dice_vs_golden(case):
    """dice_loss_sa / logcoshdiceloss (reference loss/dice.py:275-349) on the fused kernels: values and dlogits against
    the vectors captured from the reference (tests/golden/losses_aux.npz)."""
    import os
    from loss.dice import dice_loss_sa, logcoshdiceloss
    name, C, N, side, kw = case
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "losses_aux.npz"))
    z, t = _inputs(name, C, N, side, kw)
    nos = bool(kw.get("no_sft"))
    zin = torch.softmax(z, 1) if nos else z
    for tag, fn in (("sa", dice_loss_sa), ("logcosh", logcoshdiceloss)):
        a = zin.clone().cuda().requires_grad_(True)
        loss = fn(a, t.cuda(), no_sft=nos)
        loss.backward()
        report(f"{name}/{tag}", loss.detach().cpu(), torch.from_numpy(g[f"{name}/{tag}"]), rtol=1e-5, atol=1e-6)
        want = torch.from_numpy(g[f"{name}/{tag}_grad"])
        report(f"{name}/{tag}_grad", a.grad.cpu(), want, rtol=1e-4, atol=1e-4 * float(want.abs().max()) + 1e-12)


def test_edge_dice_loss_vs_golden():
    """edge_conv_loss / get_edg_c (reference loss/hausdorff.py:71-128) on the HIP edge kernels + the fused Dice kernels:
    edge map, loss and dlogits against the vectors captured from the reference (tests/golden/losses_edge.npz)."""
    import os
    from oracle.cases import loss_cases
    from loss.edge_loss import get_hot_enc
    from loss.hausdorff import edge_conv_loss, get_edg_c
    from segk import _lib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "losses_edge.npz"))
    for name, C, N, side, kw in loss_cases():
        if f"{name}/edge_loss" not in g.files:
            continue
        z, t = _inputs(name, C, N, side, kw)
        a = z.clone().cuda().requires_grad_(True)
        before = _lib.launches
        report(f"{name}/edge_map", get_edg_c(a).detach().cpu(), torch.from_numpy(g[f"{name}/edge_map"]), rtol=1e-4, atol=1e-5)
        report(f"{name}/edge_target", get_edg_c(get_hot_enc(t.cuda())).cpu(), torch.from_numpy(g[f"{name}/edge_target"]), rtol=1e-4, atol=1e-5)
        loss = edge_conv_loss(a, t.cuda())
        loss.backward()
        assert _lib.launches - before >= 6, "the HIP library did not run the loss"
        report(f"{name}/edge_loss", loss.detach().cpu(), torch.from_numpy(g[f"{name}/edge_loss"]), rtol=1e-5, atol=1e-6)
        want = torch.from_numpy(g[f"{name}/edge_grad"])
        report(f"{name}/edge_grad", a.grad.cpu(), want, rtol=1e-4, atol=1e-4 * float(want.abs().max()))


def test_edge_map_operator_vs_oracle():
    """The edge operator alone on a non-square, multi-plane input with an arbitrary upstream gradient, against the oracle
    in fp64; two runs give identical bits."""
    from oracle import ref_losses as RL
    from loss.hausdorff import get_edg_c
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(3, 5, 37, 50, generator=gen)
    up = torch.randn(3, 5, 37, 50, generator=gen)
    a = x.clone().double().requires_grad_(True)
    # fp64 oracle: the restatement builds fp32 filters, so evaluate it through torch's double kernels explicitly
    import torch.nn.functional as F
    kx = torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]], dtype=torch.float64)
    ky = torch.tensor([[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]], dtype=torch.float64)
    want = sum(torch.relu(F.instance_norm(F.conv2d(a, k.expand(5, 1, 3, 3), padding=1, groups=5), eps=1e-5)) for k in (kx, -kx, ky, -ky))
    want.backward(up.double())
    outs = []
    for _ in range(2):
        b = x.clone().cuda().requires_grad_(True)
        e = get_edg_c(b)
        e.backward(up.cuda())
        outs.append((e.detach().clone(), b.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    report("edge_map", outs[0][0].cpu(), want.detach().float(), rtol=1e-4, atol=1e-5)
    report("edge_dx", outs[0][1].cpu(), a.grad.float(), rtol=1e-4, atol=1e-4 * float(a.grad.abs().max()))
    # fp32 oracle agrees with its fp64 evaluation to the same tolerance (sanity of the comparison itself)
    report("oracle32", RL.get_edg_c(x), want.detach().float(), rtol=1e-4, atol=1e-5)


def test_unsquared_dice_metric_vs_golden():
    """dice_coefficientold (reference loss/dice.py:186-256) from two passes of the fused forward kernel, against the
    reference's vectors (tests/golden/losses_misc.npz)."""
    import os
    from oracle.cases import loss_cases
    from loss.dice import dice_coefficientold
    from segk import _lib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "losses_misc.npz"))
    for name, C, N, side, kw in loss_cases():
        if f"{name}/dice_old" not in g.files:
            continue
        z, t = _inputs(name, C, N, side, kw)
        before = _lib.launches
        got = dice_coefficientold(z.cuda(), t.cuda())
        assert _lib.launches - before == 2
        report(f"{name}/dice_old", got.cpu(), torch.from_numpy(g[f"{name}/dice_old"]), rtol=1e-5, atol=1e-6)
        report(f"{name}/dice_old_list", dice_coefficientold(z.cuda(), t.cuda(), is_list_bat=True).cpu(),
               torch.from_numpy(g[f"{name}/dice_old_list"]), rtol=1e-5, atol=1e-6)
        probs = torch.softmax(z, 1)
        report(f"{name}/dice_old_nosft", dice_coefficientold(probs.cuda(), t.cuda(), nosft=True).cpu(), torch.from_numpy(g[f"{name}/dice_old"]),
               rtol=1e-5, atol=1e-6)


def test_dice_plus_focal_run_one_forward_and_one_backward_pass():
    """`dice_loss_sq(out, t) + focal_loss_with_logits(out, t)` (the training loops' loss) shares one autograd node: two
    C-ABI calls forward (sums + finalize), one backward, and the same values and gradient as the two losses taken apart."""
    import torch
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from segk import _lib
    g = torch.Generator().manual_seed(5)
    z = torch.randn(3, 4, 48, 48, generator=g).cuda().requires_grad_(True)
    t = torch.randint(0, 4, (3, 48, 48), generator=g).cuda()
    before = _lib.launches
    d, f = dice_loss_sq(z, t), focal_loss_with_logits(z, t)
    (d + f).backward()
    assert _lib.launches - before == 3, _lib.launches - before
    apart = []
    for fn in (dice_loss_sq, focal_loss_with_logits):
        z2 = z.detach().clone().requires_grad_(True)
        v = fn(z2, t)
        v.backward()
        apart.append((v.detach(), z2.grad))
    assert float(apart[0][0]) == float(d) and float(apart[1][0]) == float(f)
    want = apart[0][1] + apart[1][1]
    assert float((z.grad - want).abs().max()) <= 1e-7 + 1e-5 * float(want.abs().max())
    # a node that has run its backward is not handed out again
    before = _lib.launches
    focal_loss_with_logits(z, t).backward()
    assert _lib.launches - before == 3
