"""Full-network parity on the GPU: logits / loss / parameter gradients / BatchNorm buffers / arg-max masks of
the HIP path against the golden vectors captured from the reference (tests/golden/models.npz, variants.npz).

fp32 path: rel <= 1e-4 (north-star bound) on logits and loss, masks bit-exact away from near-ties;
bf16 path: rel <= 2e-2 (relative L2 per tensor), masks compared where the fp32 top-2 margin exceeds the
bf16 logit error.
"""
import numpy as np
import pytest
import torch

from util import VARIANTS, input_channels, rel_l2, report, split_outputs, step_loss, variant_oracle, variant_product

pytestmark = pytest.mark.gpu


def build(name):
    from models.unetvgg import UNetVgg, UNetVggInOut
    if name == "unetvgg_c4":
        return UNetVgg(out_classes=4), 3, 4
    if name == "unetvgg_c3":
        return UNetVgg(out_classes=3), 3, 3
    if name == "unetvgg_c4_convtrans":
        return UNetVgg(out_classes=4, if_convtrans=True), 3, 4
    if name == "unetvgginout_c4":
        return UNetVggInOut(in_classes=1, out_classes=4), 1, 4
    if name == "nestedunet_c4":
        from models.unetplusplus import Nested_UNet
        return Nested_UNet(in_ch=1, out_ch=4), 1, 4
    if name in VARIANTS:
        return variant_product(name), 1, 3 if name.endswith("c3") else 4
    raise KeyError(name)


def run_step(name, dtype, side=64, n=2, tag=None):
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from oracle.ref_models import hash_init_, synthetic_batch
    from segk import _lib
    model, in_ch, n_cls = build(name)
    hash_init_(model)
    model = model.cuda().train().set_compute_dtype(dtype)
    x, t = synthetic_batch(n, side, n_cls, in_ch=in_ch, tag=tag or f"model:{name}")
    before = _lib.launches
    logits, feats, loss = step_loss(model(x.cuda()), t.cuda(), dice_loss_sq, focal_loss_with_logits)
    loss.backward()
    torch.cuda.synchronize()
    assert _lib.launches - before > 50, "the HIP library did not run the network"
    run_step.feats = [f.detach() for f in feats]
    return model, logits.detach(), loss.detach()


MODELS = ["unetvgg_c4", "unetvgg_c3", "unetvgginout_c4", "nestedunet_c4", "unetvgg_c4_convtrans"] + list(VARIANTS)


def oracle_step(name, dtype, side=64, n=2, tag=None, autocast=False):
    """The CPU oracle (pinned bit for bit to the reference) on the same inputs, in fp32 or fp64."""
    from oracle import ref_losses as RL
    from oracle import ref_models as RM
    mk = {"unetvgg_c4": lambda: RM.RefUNetVgg(out_classes=4), "unetvgg_c3": lambda: RM.RefUNetVgg(out_classes=3),
          "unetvgg_c4_convtrans": lambda: RM.RefUNetVgg(out_classes=4, if_convtrans=True),
          "unetvgginout_c4": lambda: RM.RefUNetVgg(out_classes=4, in_classes=1),
          "nestedunet_c4": lambda: RM.RefNestedUNet(in_ch=1, out_ch=4)}.get(name) or (lambda: variant_oracle(name))
    in_ch = input_channels(name)
    n_cls = 3 if name.endswith("c3") else 4
    x, t = RM.synthetic_batch(n, side, n_cls, in_ch=in_ch, tag=tag or f"model:{name}")
    m = RM.hash_init_(mk()).to(dtype).train()
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            logits, _ = split_outputs(m(x))
        return m, logits.detach().float(), None
    logits, _, loss = step_loss(m(x.to(dtype)), t, RL.dice_loss_sq, RL.focal_loss_with_logits)
    loss.backward()
    return m, logits.detach(), loss.detach()


def check_fp32_grads(name, model, side=64, n=2, tag=None):
    """End-to-end fp32 gradients against the fp64 oracle.

    A bound of 1e-4 is only meaningful where the network is well conditioned.  Measured on the CPU oracle: the
    reference's own fp32 gradients are 3e-4 .. 1.5e-3 (relative L2) away from exact arithmetic, and an fp64 run
    whose INPUT is perturbed by 1e-6 (which moves the logits by 2e-5, the size of any fp32 forward error)
    changes the gradients by 2e-3 .. 1e-2 in discrete jumps: ReLU masks flip at the few high-leverage pixels
    that carry the Dice/focal gradient.  Hence: layers behind the last such flip (final, dec1.0) must meet
    1e-4; every tensor must stay within 2.5e-2, or within 1.5x the reference's own fp32 error where that is larger;
    the per-kernel backward passes are held to 1e-4 separately (tests/test_gpu_ops.py)."""
    m32, _, _ = oracle_step(name, torch.float32, side, n, tag)
    m64, _, _ = oracle_step(name, torch.float64, side, n, tag)
    rows = []
    for (k, p), (_, q32), (_, q64) in zip(model.named_parameters(), m32.named_parameters(), m64.named_parameters()):
        g64 = q64.grad
        if g64 is None:       # a parameter the forward never uses (SpiderNet.conv5)
            assert p.grad is None or float(p.grad.abs().max()) == 0, k
            continue
        scale = float(g64.norm())
        if scale < 1e-6:      # analytically zero gradients (conv bias in front of a batch-statistics norm)
            assert float(p.grad.norm()) < 1e-5, k
            continue
        e_ref = float((q32.grad.double() - g64).norm()) / scale
        e_hip = float((p.grad.double().cpu() - g64).norm()) / scale
        rows.append((e_hip, e_ref, k))
        bound = 1e-4 if k.startswith(("final.", "dec1.0.")) else max(2.5e-2, 1.5 * e_ref)
        assert e_hip <= bound, f"{k}: rel-L2 vs fp64 oracle {e_hip:.3e} (reference fp32: {e_ref:.3e}), bound {bound}"
    return max(rows)


@pytest.mark.parametrize("name", MODELS)
def test_fp32_step_vs_golden(name, golden):
    g = golden["models"]
    model, logits, loss = run_step(name, torch.float32)
    want = torch.from_numpy(g[f"{name}/logits"])
    report(f"{name}/logits", logits.cpu(), want, rtol=1e-4, atol=1e-4)
    report(f"{name}/loss", loss.cpu(), torch.from_numpy(g[f"{name}/loss"]), rtol=1e-4, atol=1e-6)
    for i, f in enumerate(run_step.feats):       # extra outputs of the multi-output networks
        report(f"{name}/feat{i}", f[..., ::4, ::4].cpu(), torch.from_numpy(g[f"{name}/feat{i}_sub"]), rtol=1e-4, atol=1e-4)
    # arg-max masks: bit-exact wherever the reference's top-2 margin exceeds the logit tolerance
    top2 = want.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 4e-4
    mask = logits.argmax(1).to(torch.uint8).cpu()
    assert torch.equal(mask[safe], torch.from_numpy(g[f"{name}/mask"])[safe])
    assert int((~safe).sum()) <= 4 * int(g[f"{name}/near_tie_1e-3"]) + 8
    # gradients: golden norms within the reference's own noise band, full tensors against the fp64 oracle
    for k, p in model.named_parameters():
        if f"{name}/grad_norm/{k}" not in g.keys():      # never used by the reference's forward: no gradient there either
            assert p.grad is None or float(p.grad.abs().max()) == 0, k
            continue
        assert p.grad is not None, k
        gn = float(g[f"{name}/grad_norm/{k}"])
        if gn > 1e-6:   # loose band: the reference's own fp32 noise on these norms reaches ~1e-2 on tiny tensors
            assert abs(float(p.grad.double().norm()) - gn) / gn < 2e-2, f"{k}: grad norm {float(p.grad.norm())} vs {gn}"
            # the first 16 values of the reference's gradient, within the same band relative to the tensor's norm
            head = torch.from_numpy(g[f"{name}/grad_head/{k}"]).double()
            got = p.grad.detach().flatten()[:head.numel()].double().cpu()
            assert float((got - head).norm()) <= 2.5e-2 * gn, f"{k}: first gradient values {got[:4].tolist()} vs {head[:4].tolist()}"
    worst = check_fp32_grads(name, model)
    print(f"{name}: worst gradient rel-L2 vs fp64 oracle (hip, reference fp32, name) = {worst}")
    for k, b in model.named_buffers():
        if k.endswith("num_batches_tracked"):
            assert int(b) == 1
        else:
            report(f"{name}/buf/{k}", b.cpu(), torch.from_numpy(g[f"{name}/buf/{k}"]), rtol=1e-4, atol=1e-5)
    # eval mode uses the running buffers just updated
    model.eval()
    from oracle.ref_models import synthetic_batch
    x, _ = synthetic_batch(2, 64, 3 if name.endswith("c3") else 4, in_ch=input_channels(name), tag=f"model:{name}")
    with torch.no_grad():
        le, _ = split_outputs(model(x.cuda()))
    report(f"{name}/eval_logits", le[..., ::4, ::4].cpu(), torch.from_numpy(g[f"{name}/eval_logits_sub"]), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", ["unetvgg_c4", "unetsimple_c4", "nestedunet_c4", "unetvgg_c4_convtrans"])
def test_bf16_step_vs_golden(name, golden):
    """bf16 path.  Every kernel is within rel 2e-2 of its fp32 op (tests/test_gpu_ops.py); end to end, 25 stacked
    layers accumulate more than that in ANY bf16 implementation, so the network-level bound is the error of
    the oracle itself run under torch bf16 autocast on the CPU (same rounding points), with 5e-2 as a cap."""
    g = golden["models"]
    model, logits, loss = run_step(name, torch.bfloat16)
    want = torch.from_numpy(g[f"{name}/logits"])
    _, auto, _ = oracle_step(name, torch.float32, autocast=True)
    e_ref = rel_l2(auto, want)
    e_hip = rel_l2(logits, want)
    print(f"{name}: bf16 logits rel-L2 hip {e_hip:.3e}, torch-autocast oracle {e_ref:.3e}")
    assert e_hip < min(5e-2, 1.25 * e_ref + 2e-3), (e_hip, e_ref)
    assert abs(float(loss) - float(g[f"{name}/loss"])) < 2e-2 * float(g[f"{name}/loss"])
    err = float((logits.cpu() - want).abs().max())
    top2 = want.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2 * err
    assert torch.equal(logits.argmax(1).to(torch.uint8).cpu()[safe], torch.from_numpy(g[f"{name}/mask"])[safe])
    assert float(safe.float().mean()) > 0.3
    # gradients: the same network under torch's bf16 autocast is 1e-2 (last layers) .. 6e-1 (first layers) away
    # from exact arithmetic on this ill-conditioned random-weight case; the HIP path must not be worse
    m64, _, _ = oracle_step(name, torch.float64)
    from oracle import ref_losses as RL
    mauto, _, _ = oracle_step(name, torch.float32, autocast=True)
    from oracle.ref_models import synthetic_batch
    x, t = synthetic_batch(2, 64, 4, in_ch=input_channels(name), tag=f"model:{name}")
    with torch.autocast("cpu", dtype=torch.bfloat16):
        la = mauto(x)
    (RL.dice_loss_sq(la.float(), t) + RL.focal_loss_with_logits(la.float(), t)).backward()
    bad = []
    for (k, p), (_, q), (_, qa) in zip(model.named_parameters(), m64.named_parameters(), mauto.named_parameters()):
        scale = float(q.grad.norm())
        if scale < 1e-6:
            continue
        e = float((p.grad.double().cpu() - q.grad).norm()) / scale
        ea = float((qa.grad.double() - q.grad).norm()) / scale
        if e > 1.3 * ea + 2e-2:
            bad.append((k, e, ea))
    assert not bad, bad


def test_unetvgg384_fp32_mask(golden):
    """One full-size slice (N=1, 384x384): mask, sub-sampled logits, loss and gradient norms."""
    g = golden["models"]
    model, logits, loss = run_step("unetvgg_c4", torch.float32, side=384, n=1, tag="model:unetvgg384")
    want = torch.from_numpy(g["unetvgg384/logits_sub"])
    report("unetvgg384/logits_sub", logits[..., ::8, ::8].cpu(), want, rtol=1e-4, atol=1e-4)
    report("unetvgg384/loss", loss.cpu(), torch.from_numpy(g["unetvgg384/loss"]), rtol=1e-4, atol=1e-6)
    mask = logits.argmax(1).to(torch.uint8).cpu()
    ref = torch.from_numpy(g["unetvgg384/mask"])
    # logits agree to 1e-4 (checked above on the sub-sampled grid), so wherever the top-2 margin of the HIP logits
    # exceeds 4e-4 the arg-max is decided: there every pixel must equal the reference mask; elsewhere at most the
    # near-tie pixels may differ
    top2 = logits.cpu().topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 4e-4
    assert torch.equal(mask[safe], ref[safe]), f"{int((mask[safe] != ref[safe]).sum())} decided pixels differ"
    diff = int((mask != ref).sum())
    print(f"unetvgg384: {diff} of {mask.numel()} mask pixels differ; {int((~safe).sum())} pixels have a margin below 4e-4")
    assert diff <= int((~safe).sum()) and diff <= int(g["unetvgg384/near_tie_1e-3"]), f"{diff} mask pixels differ"
    for k, p in model.named_parameters():
        gn = float(g[f"unetvgg384/grad_norm/{k}"])
        if gn > 1e-6:     # band = the reference's own fp32 noise on these norms (see check_grads_at_reference_noise_floor)
            assert abs(float(p.grad.double().norm()) - gn) / gn < 5e-3, k


@pytest.mark.parametrize("tag", ["wellcond:8", "wellcond-convtrans:5", "wellcond-unetpp:8"])
def test_bf16_meets_2e2_on_well_conditioned_cases(tag):
    """The bf16 path against the north-star bound (rel <= 2e-2) where it can hold: on the well-conditioned cases the logits and
    the loss of the HIP bf16 path are within 2e-2 of the fp32 reference (measured 3e-3 .. 7e-3 and <= 2e-3).  Parameter gradients
    are not: the same network under torch's bf16 autocast on the CPU (the same rounding points) is 2e-2 .. 2e-1 away from the fp32
    gradients on these cases, so no bf16 implementation meets 2e-2 there; the HIP gradients must be no further away than that
    autocast run, tensor by tensor and in the median (measured: HIP median 2.3e-2 .. 4.6e-2, autocast 2.5e-2 .. 6.9e-2)."""
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from oracle import ref_losses as RL
    from oracle import wellcond as WC
    x, t = WC.batch(tag)
    ref = WC.oracle_model(tag)
    lr = ref(x)
    loss_r = RL.dice_loss_sq(lr, t) + RL.focal_loss_with_logits(lr, t)
    loss_r.backward()
    auto = WC.oracle_model(tag)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        la = auto(x)
    (RL.dice_loss_sq(la.float(), t) + RL.focal_loss_with_logits(la.float(), t)).backward()
    net = WC.hip_model(tag).cuda().set_compute_dtype(torch.bfloat16)
    lg = net(x.cuda())
    loss = dice_loss_sq(lg, t.cuda()) + focal_loss_with_logits(lg, t.cuda())
    loss.backward()
    torch.cuda.synchronize()
    e_logits = rel_l2(lg.detach(), lr.detach())
    e_loss = abs(float(loss) - float(loss_r)) / abs(float(loss_r))
    assert e_logits <= 2e-2 and e_loss <= 2e-2, (e_logits, e_loss)
    eh, ea = [], []
    for (k, p), (_, q), (_, qa) in zip(net.named_parameters(), ref.named_parameters(), auto.named_parameters()):
        scale = float(q.grad.norm())
        if scale < 1e-6:
            continue
        e = float((p.grad.double().cpu() - q.grad.double()).norm()) / scale
        a = float((qa.grad.double() - q.grad.double()).norm()) / scale
        assert e <= 1.3 * a + 2e-2, (k, e, a)
        eh.append(e)
        ea.append(a)
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"{tag}: bf16 logits rel-L2 {e_logits:.2e}, loss {e_loss:.2e}; gradient rel-L2 median {med(eh):.2e} worst {max(eh):.2e} "
          f"(torch bf16 autocast oracle: median {med(ea):.2e} worst {max(ea):.2e})")
    assert med(eh) <= 1.1 * med(ea) + 5e-3 and max(eh) <= 1.2 * max(ea) + 2e-2


def _bench_shape_properties(model, x, t, loss_fn):
    """Two training steps from the same state give identical bits (logits and every gradient: fixed-order reductions everywhere),
    everything is finite, and in eval mode an image's logits do not depend on the rest of the batch."""
    heads = lambda out: list(out) if isinstance(out, (list, tuple)) else [out]
    outs = []
    for _ in range(2):
        model.zero_grad()
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        logits = model(x)
        loss = loss_fn(logits, t)
        loss.backward()
        outs.append(([h.detach().clone() for h in heads(logits)], [p.grad.clone() for p in model.parameters()], float(loss)))
    for a, b in zip(outs[0][0], outs[1][0]):
        assert torch.equal(a, b) and torch.isfinite(a).all()
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b) and torch.isfinite(a).all()
    assert outs[0][2] == outs[1][2] and np.isfinite(outs[0][2])
    model.eval()
    with torch.no_grad():
        full = heads(model(x))
        part = heads(model(x[2:4]))
    for f, p in zip(full, part):
        assert torch.equal(f[2:4], p)


def test_bench_shape_properties():
    """BASELINE configs[1] at its full size (UNet-VGG, batch 32, 384x384, bf16): size-independent properties instead of an
    oracle run."""
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from models.unetvgg import UNetVgg
    from oracle.ref_models import hash_init_, synthetic_batch
    model = hash_init_(UNetVgg(out_classes=4)).cuda().train().set_compute_dtype(torch.bfloat16)
    x, t = synthetic_batch(32, 384, 4, tag="bench")
    _bench_shape_properties(model, x.cuda(), t.cuda(), lambda lg, tt: dice_loss_sq(lg, tt) + focal_loss_with_logits(lg, tt))


def test_bench_shape_properties_unetpp_deep_supervision():
    """BASELINE configs[3] at its full size (UNet++ with four supervised heads, batch 16, 384x384, bf16, summed Dice + focal):
    the same size-independent properties, on every head."""
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from models.unetplusplus import Nested_UNet, deep_supervision_loss
    from oracle.ref_models import hash_init_, synthetic_batch
    model = hash_init_(Nested_UNet(in_ch=1, out_ch=4, deep_supervision=True)).cuda().train().set_compute_dtype(torch.bfloat16)
    x, t = synthetic_batch(16, 384, 4, in_ch=1, tag="bench-unetpp")
    _bench_shape_properties(model, x.cuda(), t.cuda(),
                            lambda lg, tt: deep_supervision_loss(lg, tt, [dice_loss_sq, focal_loss_with_logits]))


BLOCKS = [
    ("decoder_bilinear", dict(if_convtrans=False), (2, 64, 16, 24)),
    ("decoder_convtrans", dict(if_convtrans=True), (2, 64, 16, 24)),
    ("decoder_noupsam", dict(noupsam=True), (2, 64, 16, 24)),
    ("double_conv", None, (2, 32, 24, 16)),
]


@pytest.mark.parametrize("kind,kw,shape", BLOCKS, ids=[b[0] for b in BLOCKS])
def test_standalone_blocks(kind, kw, shape):
    """make_decoder_block / conv_block_nested used on their own on CUDA tensors (models/unetvgg.py:6-41,
    models/unetplusplus.py:6-24): output, input gradient and parameter gradients against the oracle block."""
    from oracle import ref_models as RM
    from segk import _lib
    N, C, H, W = shape
    if kw is None:
        from models.unetplusplus import conv_block_nested
        blk, ref = conv_block_nested(C, 48, 40), RM.RefDoubleConv(C, 48, 40)
    else:
        from models.unetvgg import make_decoder_block
        blk, ref = make_decoder_block(C, 48, 40, **kw), RM.decoder_block(C, 48, 40, **kw)
    RM.hash_init_(blk), RM.hash_init_(ref)
    ref = ref.double().train()
    blk = blk.cuda().train()
    x = RM.hash_uniform(f"block:{kind}:x", shape)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    dy = RM.hash_uniform(f"block:{kind}:dy", tuple(yr.shape))
    yr.backward(dy.double())
    xg = x.cuda().requires_grad_(True)
    before = _lib.launches
    y = blk(xg)
    y.backward(dy.cuda())
    assert _lib.launches - before >= 6
    report(f"{kind}/out", y.detach().cpu(), yr.detach().float(), rtol=1e-4, atol=1e-4)
    assert rel_l2(xg.grad.cpu(), xr.grad.float()) < 1e-4
    for (k, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        scale = float(q.grad.norm())
        if scale < 1e-9:
            assert float(p.grad.norm()) < 1e-5, k
        else:
            assert float((p.grad.double().cpu() - q.grad).norm()) / scale < 1e-4, k
    # the CPU path of the same module is plain torch.nn
    yc = blk.cpu()(x)
    report(f"{kind}/host_out", yc.detach(), yr.detach().float(), rtol=1e-4, atol=1e-5)


def test_weight_gradients_on_side_stream(monkeypatch):
    """Opt-in SEGK_WGRAD_STREAM=1: weight-gradient kernels run on a second stream next to the data gradients.  Same
    kernels, same order per tensor, so every gradient is bitwise identical to the single-stream run."""
    from segk import engine
    grads = []
    for flag in (False, True):
        monkeypatch.setattr(engine, "_WGRAD_STREAM", flag)
        model, logits, loss = run_step("unetvgg_c4", torch.bfloat16)
        grads.append([p.grad.clone() for p in model.parameters()] + [logits, loss])
    for a, b in zip(*grads):
        assert torch.equal(a, b)


@pytest.mark.parametrize("name", ["unetvgg_c4", "nestedunet_c4"])
def test_norm_backward_sums_from_the_data_gradient(monkeypatch, name):
    """Opt-in SEGK_NORM_FUSE=1: the data gradient of the convolution behind a BatchNorm + ReLU also reduces the two sums
    of that norm's backward.  Same arithmetic per element, sums added in another order: gradients agree to fp32 rounding."""
    from segk import engine, _lib
    grads, fused = [], []
    for flag in (False, True):
        monkeypatch.setattr(engine, "_NORM_FUSE", flag)
        before = _lib.calls.get("segk_conv_dgrad_norm", 0)
        model, logits, loss = run_step(name, torch.float32)
        fused.append(_lib.calls.get("segk_conv_dgrad_norm", 0) - before)
        grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    assert fused[0] == 0 and fused[1] >= 8, f"fused data gradients launched: {fused}"
    for k, a in grads[0].items():
        b = grads[1][k]
        scale = float(a.norm())
        assert float((a - b).norm()) <= 2e-5 * scale + 1e-9, k


def test_unetsimple_reference_width_vs_oracle():
    """UNetSimple as the reference trains it (roi_train/train_roi.py:39,52: one input channel, channelscale 128, up to
    1024-wide layers and a 2048-channel concat): fp32 logits, loss and the last layers' gradients against the oracle on
    the same weights; bf16 logits within the bound of the bf16 network test."""
    from loss.dice import dice_loss_sq
    from models.unetvgg import UNetSimple
    from oracle import ref_losses as RL
    from oracle.ref_models import RefUNetSimple, hash_init_, synthetic_batch
    x, t = synthetic_batch(2, 96, 2, in_ch=1, tag="model:unetsimple128")
    ref = hash_init_(RefUNetSimple(out_classes=2, channelscale=128)).train()
    want = ref(x)
    lw = RL.dice_loss_sq(want, t)
    lw.backward()
    model = hash_init_(UNetSimple(in_classes=1, channelscale=128, out_classes=2)).cuda().train().set_compute_dtype(torch.float32)
    got = model(x.cuda())
    loss = dice_loss_sq(got, t.cuda())
    loss.backward()
    report("unetsimple128/logits", got.detach().cpu(), want.detach(), rtol=1e-4, atol=1e-4)
    report("unetsimple128/loss", loss.detach().cpu(), lw.detach(), rtol=1e-4, atol=1e-6)
    for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if k.startswith(("final.", "dec1.0.")):
            assert rel_l2(p.grad.cpu(), q.grad) < 1e-4, k
        elif float(q.grad.norm()) > 1e-6:
            assert rel_l2(p.grad.cpu(), q.grad) < 5e-2, (k, rel_l2(p.grad.cpu(), q.grad))
    model.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        e = rel_l2(model(x.cuda()).cpu(), want.detach())
    assert e < 5e-2, e


def test_spidernet_flat_adam_and_bf16():
    """SpiderNet as perlosstrain/train_percloss.py:39 builds it (multiout): a step through the flat parameter / gradient
    buffers with the fused Adam (its never-used conv5 parameters get a zero gradient there), and bf16 logits against the
    fp32 oracle."""
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from models.unetvgg import SpiderNet
    from oracle.ref_models import RefSpiderNet, hash_init_, synthetic_batch
    from segk.flat import FlatModel
    from segk.optim import Adam
    x, t = synthetic_batch(2, 64, 3, in_ch=1, tag="model:spidernet_flat")
    ref = hash_init_(RefSpiderNet(out_classes=3, channelscale=32, multiout=True)).train()
    want, _ = ref(x)
    model = hash_init_(SpiderNet(out_classes=3, channelscale=32, multiout=True)).cuda().train().set_compute_dtype(torch.bfloat16)
    flat = FlatModel(model)
    opt = Adam(flat, lr=1e-3)
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    logits, feats, loss = step_loss(model(x.cuda()), t.cuda(), dice_loss_sq, focal_loss_with_logits)
    assert rel_l2(logits, want) < 5e-2 and len(feats) == 5
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    for k, p in model.named_parameters():
        moved = float((p.detach() - before[k]).abs().max())
        if k.startswith("conv5."):
            assert moved == 0.0 and float(p.grad.abs().max()) == 0.0, k
        else:
            assert torch.isfinite(p).all() and (moved > 0.0 or float(p.grad.abs().max()) < 1e-12), k


@pytest.mark.parametrize("name", ["unetvgg256", "unetvggtwo"])
def test_resizing_wrappers_vs_golden(name):
    """UNetVgg256 / UNetVggTwo (models/unetvgg.py:107-132, 216-265): torch re-sizing at the boundary around HIP networks;
    logits, loss and gradient norms of one training step against the reference's vectors (tests/golden/wrappers.npz)."""
    import os
    import models.unetvgg as M
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from oracle.ref_models import hash_init_, synthetic_batch
    from segk import _lib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wrappers.npz"))
    two = name == "unetvggtwo"
    x, t = synthetic_batch(2, 96, 3, in_ch=1 if two else 3, tag=f"model:{name}")
    model = hash_init_(M.UNetVggTwo() if two else M.UNetVgg256()).cuda().train()
    for m in model.children():
        m.set_compute_dtype(torch.float32)
    before = _lib.launches
    y = model(x.cuda(), t.cuda()) if two else model(x.cuda())
    loss = dice_loss_sq(y, t.cuda()) + focal_loss_with_logits(y, t.cuda())
    loss.backward()
    assert _lib.launches - before > 100, "the HIP library did not run the networks"
    report(f"{name}/logits", y.detach().cpu(), torch.from_numpy(g[f"{name}/logits"]), rtol=1e-4, atol=1e-4)
    report(f"{name}/loss", loss.detach().cpu(), torch.from_numpy(g[f"{name}/loss"]), rtol=1e-4, atol=1e-6)
    for k, p in model.named_parameters():
        gn = float(g[f"{name}/grad_norm/{k}"])
        if gn > 1e-6:
            assert abs(float(p.grad.double().norm()) - gn) / gn < 2e-2, f"{k}: grad norm {float(p.grad.norm())} vs {gn}"


def _deep_sup_pair(dtype):
    """UNet++ with four supervised heads (an extension: the reference has one head, so the oracle for it is this
    repository's own torch restatement, NOT pinned by the reference), summed Dice + focal over the heads."""
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from models.unetplusplus import Nested_UNet, deep_supervision_loss
    from oracle import ref_losses as RL
    from oracle import ref_models as RM
    x, t = RM.synthetic_batch(2, 64, 4, in_ch=1, tag="model:nestedunet_deepsup")
    ref = RM.hash_init_(RM.RefNestedUNet(in_ch=1, out_ch=4, deep_supervision=True)).double().train()
    outs_r = ref(x.double())
    loss_r = deep_supervision_loss(outs_r, t, [RL.dice_loss_sq, RL.focal_loss_with_logits])
    loss_r.backward()
    net = RM.hash_init_(Nested_UNet(in_ch=1, out_ch=4, deep_supervision=True)).cuda().train().set_compute_dtype(dtype)
    outs = net(x.cuda())
    loss = deep_supervision_loss(outs, t.cuda(), [dice_loss_sq, focal_loss_with_logits])
    loss.backward()
    torch.cuda.synchronize()
    return net, outs, loss, ref, outs_r, loss_r


def test_nestedunet_deep_supervision_fp32():
    net, outs, loss, ref, outs_r, loss_r = _deep_sup_pair(torch.float32)
    assert isinstance(outs, list) and len(outs) == 4
    for j, (o, r) in enumerate(zip(outs, outs_r)):
        report(f"deepsup/head{j + 1}", o.detach().cpu(), r.detach(), rtol=1e-4, atol=1e-4)
    assert abs(float(loss) - float(loss_r)) < 1e-4 * max(1.0, abs(float(loss_r)))
    worst = 0.0
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        scale = float(q.grad.norm())
        if scale < 1e-6:
            assert float(p.grad.norm()) < 1e-5, k
            continue
        e = float((p.grad.double().cpu() - q.grad).norm()) / scale
        worst = max(worst, e)
        # heads sit behind the last ReLU: 1e-4; deeper tensors carry the ReLU-mask flips of a random-weight network
        assert e <= (1e-4 if k.startswith("final") else 2.5e-2), (k, e)
    print(f"deep supervision fp32: worst gradient rel-L2 vs fp64 oracle {worst:.3e}")


def test_nestedunet_deep_supervision_bf16():
    net, outs, loss, ref, outs_r, loss_r = _deep_sup_pair(torch.bfloat16)
    for j, (o, r) in enumerate(zip(outs, outs_r)):
        e = rel_l2(o, r)
        assert e < 5e-2, (j, e)          # network-level bf16 bound, as for the single-head network
    assert abs(float(loss) - float(loss_r)) < 2e-2 * abs(float(loss_r))
    for k in ("final1", "final2", "final3", "final4"):
        p, q = dict(net.named_parameters())[k + ".weight"], dict(ref.named_parameters())[k + ".weight"]
        assert rel_l2(p.grad, q.grad) < 1e-1, k


@pytest.mark.parametrize("tag", ["wellcond:8", "wellcond:7", "wellcond-convtrans:5", "wellcond-unetpp:8"])
def test_fp32_gradients_meet_1e4_on_well_conditioned_cases(tag):
    """End-to-end fp32 gradient parity at the north-star bound, for UNetVgg (bilinear decoder, two batches), UNetVgg with the
    transposed-convolution decoder (models/unetvgg.py:31-32) and Nested_UNet (models/unetplusplus.py:27-82).  On the
    well-conditioned cases (oracle/wellcond.py: no pre-ReLU value within rounding of its threshold, so no ReLU mask depends on
    the rounding) every parameter gradient of the HIP path must be within 1e-4 (relative L2) of the REFERENCE's fp32 gradient:
    against the reference-generated fixture (norm, 4096 strided samples and four +-1 hash projections per tensor:
    tests/golden/wellcond.npz) and, tensor for tensor, against the oracle run here (pinned bit for bit to the reference on these
    cases).  Where the reference's own fp32 gradient of a tensor is more than 3e-5 from exact arithmetic (two cancelling sums of
    UNet++), two correct fp32 implementations can be further than 1e-4 apart: there the HIP gradient must be within 1e-4 of the
    fp64 oracle, and the comparison with the reference's fp32 value gets that tensor's own fp32 error added to the bound."""
    import os
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from oracle import ref_losses as RL
    from oracle import ref_models as RM
    from oracle import wellcond as WC
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wellcond.npz"))
    x, t = WC.batch(tag)
    net = WC.hip_model(tag).cuda().set_compute_dtype(torch.float32)
    lg = net(x.cuda())
    loss = dice_loss_sq(lg, t.cuda()) + focal_loss_with_logits(lg, t.cuda())
    loss.backward()
    torch.cuda.synchronize()
    report(f"{tag}/logits", lg.detach().cpu(), torch.from_numpy(g[f"{tag}/logits"]), rtol=1e-4, atol=1e-4)
    assert abs(float(loss) - float(g[f"{tag}/loss"])) <= 1e-4 * abs(float(g[f"{tag}/loss"]))
    ref = WC.oracle_model(tag)
    lr = ref(x)
    (RL.dice_loss_sq(lr, t) + RL.focal_loss_with_logits(lr, t)).backward()
    ref64 = WC.oracle_model(tag).double()
    l64 = ref64(x.double())
    (RL.dice_loss_sq(l64, t) + RL.focal_loss_with_logits(l64, t)).backward()
    worst = worst64 = 0.0
    loose = []
    for (k, p), (_, q), (_, q64) in zip(net.named_parameters(), ref.named_parameters(), ref64.named_parameters()):
        got = p.grad.detach().double().flatten().cpu()
        norm = float(g[f"{tag}/grad/{k}/norm"])
        if norm < 1e-6:       # analytically zero (a bias in front of a batch-statistics norm): rounding noise in the reference
            assert float(got.norm()) < 1e-5, k
            continue
        exact = q64.grad.flatten()
        ref_own = float((q.grad.double().flatten() - exact).norm()) / norm      # the reference's fp32 error on this tensor
        err64 = float((got - exact).norm()) / norm
        tol = 1e-4
        if ref_own > 3e-5:
            assert err64 <= 1e-4, f"{k}: rel-L2 {err64:.3e} against the fp64 oracle (the reference's fp32 gradient is {ref_own:.3e} from it)"
            tol = 1e-4 + ref_own
            loose.append((k, ref_own, err64))
        assert abs(float(got.norm()) - norm) <= tol * norm, k
        sub = torch.from_numpy(g[f"{tag}/grad/{k}/sub"]).double()
        stride = max(1, got.numel() // 4096)
        share = (sub.numel() / got.numel()) ** 0.5                     # the samples' share of the tensor's norm
        assert float((got[::stride][:4096] - sub).norm()) <= tol * norm * max(share, 1e-2) * 4 + 1e-12, k
        for i, want in enumerate(g[f"{tag}/grad/{k}/proj"]):
            sign = RM.hash_uniform(f"proj{i}:{k}", (got.numel(),), -1, 1).sign().double()
            assert abs(float(got @ sign) - float(want)) <= 4 * tol * norm, (k, i)
        err = float((got - q.grad.double().flatten()).norm()) / norm
        worst, worst64 = max(worst, err), max(worst64, err64)
        assert err <= tol, f"{k}: rel-L2 {err:.3e} against the reference's fp32 gradient"
    if loose:
        print(f"{tag}: tensors judged against the fp64 oracle (reference fp32 error, HIP error): " +
              ", ".join(f"{k} {a:.1e}/{b:.1e}" for k, a, b in loose))
    print(f"{tag}: worst gradient rel-L2 against the fp64 oracle {worst64:.2e}")
    print(f"{tag}: worst gradient rel-L2 against the reference's fp32 gradients {worst:.2e} "
          f"(reference fp32 vs fp64: {float(g[tag + '/reference_fp32_vs_fp64_worst_grad_rel_l2']):.2e}; smallest relative "
          f"pre-ReLU margin {float(g[tag + '/min_relu_margin']):.1e})")
