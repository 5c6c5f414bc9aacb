"""GPU parity of the volume kernels (3x3x3 convolution, patch gather for the strided stems, 3-D pools) against
the ATen ops the reference calls, and of the 3-D DenseNet middle-slice detector (models/resnt.py:8-45 over
models/densenet3d.py) against golden vectors captured from the reference (tests/golden/volumes.npz).

Tolerances: fp32 rel 1e-4, bf16 rel 2e-2 per kernel.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import rel_l2, report

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: dict(rtol=1e-4, atol=1e-4), torch.bfloat16: dict(rtol=2e-2, atol=2e-2)}
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "volumes.npz")


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).float()


def to_vol(x, dtype, ld=None, c0=0):
    """NCDHW fp32 (CPU) -> slice-major Act [N*D,H,W,ld] on the GPU."""
    from segk.ops import Act
    N, C, D, H, W = x.shape
    ld = ld or (C + 7) // 8 * 8
    buf = torch.zeros((N * D, H, W, ld), dtype=dtype, device="cuda")
    buf[..., c0:c0 + C] = x.permute(0, 2, 3, 4, 1).reshape(N * D, H, W, C).to(dtype).cuda()
    return Act(buf, c0, (C + 7) // 8 * 8 if c0 == 0 and ld == (C + 7) // 8 * 8 else C)


def from_vol(a, N, C=None):
    t = a.buf[..., a.c0:a.c0 + (C or a.C)].float().cpu()
    ND, H, W, Cc = t.shape
    return t.reshape(N, ND // N, H, W, Cc).permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, 4, 8, 8, 128, 32), (1, 3, 12, 20, 64, 64), (3, 1, 8, 8, 32, 32), (1, 8, 16, 16, 128, 32)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv3d(dtype, case):
    """Conv3d(3, padding 1, bias=False) forward, data gradient (+accumulate) and weight gradient, incl. depth 1
    (both neighbour taps fall outside the volume) and several volumes per batch (no leakage across volumes)."""
    from segk import ops
    N, D, H, W, Cin, Cout = case
    x = rnd((N, Cin, D, H, W), dtype, 1)
    w = rnd((Cout, Cin, 3, 3, 3), dtype, 2, scale=(2.0 / (27 * Cin)) ** 0.5)
    dy = rnd((N, Cout, D, H, W), dtype, 3)
    xa = to_vol(x, dtype, ld=Cin + 8, c0=8)
    wf, wb = ops.conv3d_pack(w.cuda(), dtype)
    out = ops.Act(torch.full((N * D, H, W, Cout + 16), 3.0, dtype=dtype, device="cuda"), 16, Cout)
    ops.conv3d_fwd(xa, D, wf, out)
    ref = F.conv3d(x.double(), w.double(), padding=1).float()
    report("conv3d_fwd", from_vol(out, N), ref, **TOL[dtype])
    assert float(out.buf[..., :16].float().min()) == 3.0      # the neighbouring channels of the concat buffer are untouched
    dya = to_vol(dy, dtype)
    dx = ops.Act.empty(N * D, H, W, Cin, dtype, "cuda")
    ops.conv3d_fwd(dya, D, wb, dx)
    ref_dx = F.conv_transpose3d(dy.double(), w.double(), padding=1).float()
    report("conv3d_dgrad", from_vol(dx, N), ref_dx, **TOL[dtype])
    ops.conv3d_fwd(dya, D, wb, dx, accumulate=True)
    t2 = {k: 2 * v for k, v in TOL[dtype].items()}
    report("conv3d_dgrad_accum", from_vol(dx, N), 2 * ref_dx, **t2)
    dw = torch.empty(Cout, Cin, 3, 3, 3, device="cuda")
    ops.conv3d_wgrad(xa, D, dya, dw)
    xr = x.double().requires_grad_(False)
    wr = w.double().requires_grad_(True)
    F.conv3d(xr, wr, padding=1).backward(dy.double())
    ref_dw = wr.grad.float()
    assert rel_l2(dw.cpu(), ref_dw) < (1e-4 if dtype == torch.float32 else 2e-2)
    report("conv3d_wgrad", dw.cpu(), ref_dw, rtol=TOL[dtype]["rtol"], atol=TOL[dtype]["atol"] * float(ref_dw.abs().max()))


STEMS = [
    # N, Cin, D, H, W, Cout, kernel, stride, pad
    (2, 1, 6, 20, 24, 3, (3, 7, 7), (1, 2, 2), (1, 3, 3)),       # models/resnt.py:30
    (1, 3, 8, 16, 16, 64, (7, 7, 7), (1, 2, 2), (3, 3, 3)),      # models/densenet3d.py:98-104
    (1, 2, 5, 9, 11, 16, (3, 3, 3), (2, 2, 1), (1, 1, 1)),       # odd extents, depth stride
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", STEMS, ids=lambda c: f"{c[1]}to{c[5]}_k{c[6][0]}")
def test_stem_conv_patches(dtype, case):
    """Strided Conv3d = im2col3d + 1x1 convolution; its data gradient = 1x1 data gradient + col2im3d."""
    from segk import ops
    N, Cin, D, H, W, Cout, k, s, p = case
    x = rnd((N, Cin, D, H, W), dtype, 1)
    w = rnd((Cout, Cin) + k, dtype, 2, scale=(2.0 / (Cin * k[0] * k[1] * k[2])) ** 0.5)
    xa = to_vol(x, dtype)
    patches, Do = ops.im2col3d(xa, D, Cin, k, s, p)
    K = Cin * k[0] * k[1] * k[2]
    cs = (Cout + 7) // 8 * 8
    w1 = torch.zeros(cs, patches.C, 1, 1)
    w1[:Cout, :K, 0, 0] = w.permute(0, 2, 3, 4, 1).reshape(Cout, K)      # tap-major, channel-minor columns
    wf, wb = ops.conv_pack(w1.cuda(), dtype)
    y = ops.Act.empty(patches.N, patches.H, patches.W, cs, dtype, "cuda")
    ops.conv_fwd(patches, wf, cs, y, taps=1)
    ref = F.conv3d(x.double(), w.double(), stride=s, padding=p).float()
    assert Do == ref.shape[2]
    report("stem_fwd", from_vol(y, N, Cout), ref, **TOL[dtype])
    dy = rnd(tuple(ref.shape), dtype, 3)
    dya = to_vol(dy, dtype)
    dpatches = ops.Act.empty(patches.N, patches.H, patches.W, patches.C, dtype, "cuda")
    ops.conv_fwd(dya, wb, patches.C, dpatches, taps=1)
    dx = ops.Act.empty(N * D, H, W, 8, dtype, "cuda")
    ops.col2im3d(dpatches, dx, D, Cin, k, s, p)
    xr = x.double().requires_grad_(True)
    F.conv3d(xr, w.double(), stride=s, padding=p).backward(dy.double())
    tol = dict(TOL[dtype])
    if dtype == torch.bfloat16:
        tol["atol"] = 2e-2 * float(xr.grad.abs().max())     # the patch gradient is rounded to bf16 before the scatter-sum
    report("stem_dgrad", from_vol(dx, N, Cin), xr.grad.float(), **tol)
    assert float(dx.buf[..., Cin:].abs().max()) == 0.0 if Cin < 8 else True


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 16, 8, 8, 8), (1, 64, 5, 7, 9), (2, 8, 1, 4, 4)], ids=lambda c: "x".join(map(str, c)))
def test_pools3d(dtype, shape):
    from segk import ops
    N, C, D, H, W = shape
    x = rnd(shape, dtype, 1)
    x[:, :, ::2] = x[:, :, ::2].clamp_min(0)                # ties (zeros after a ReLU): the first maximum must win
    xa = to_vol(x, dtype)
    # MaxPool3d(3, 2, 1)
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool3d(xr, 3, 2, 1)
    Do, Ho, Wo = ref.shape[2:]
    y = ops.Act.empty(N * Do, Ho, Wo, C, dtype, "cuda")
    arg = torch.empty(N * Do * Ho * Wo * C, dtype=torch.uint8, device="cuda")
    ops.maxpool3d_fwd(xa, D, y, arg)
    assert torch.equal(from_vol(y, N), ref.detach())
    dy = rnd(tuple(ref.shape), dtype, 2)
    ref.backward(dy)
    dx = ops.Act.empty(N * D, H, W, C, dtype, "cuda")
    ops.maxpool3d_bwd(to_vol(dy, dtype), arg, dx, D)
    report("maxpool3d_bwd", from_vol(dx, N), xr.grad, **TOL[dtype])
    # AvgPool3d(2, 2)
    if min(D, H, W) >= 2:
        xr = x.clone().requires_grad_(True)
        ref = F.avg_pool3d(xr, 2, 2)
        y = ops.Act.empty(N * (D // 2), H // 2, W // 2, C, dtype, "cuda")
        ops.avgpool3d_fwd(xa, D, y)
        report("avgpool3d_fwd", from_vol(y, N), ref.detach(), **TOL[dtype])
        dy = rnd(tuple(ref.shape), dtype, 3)
        ref.backward(dy)
        ops.avgpool3d_bwd(to_vol(dy, dtype), dx, D)
        report("avgpool3d_bwd", from_vol(dx, N), xr.grad, **TOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 1664, 2, 4, 4), (3, 40, 1, 1, 1), (1, 256, 4, 3, 5)], ids=str)
def test_global_average_pool_of_a_volume(dtype, shape):
    """F.adaptive_avg_pool3d(x, 1) of models/densenet3d.py:153-157 and its adjoint."""
    from segk import ops
    N, C, D, H, W = shape
    x = rnd(shape, dtype, 31)
    xa = to_vol(x, dtype, ld=C + 8, c0=8)
    got = ops.rows_mean_fwd(xa, N)
    report("mean", got.cpu(), x.double().mean(dim=(2, 3, 4)).float(), rtol=1e-5, atol=1e-5)
    dout = rnd((N, C), torch.float32, 32)
    dxa = ops.Act(torch.full_like(xa.buf, float("nan")), 8, C)
    ops.rows_mean_bwd(dout.cuda(), dxa)
    want = (dout / (D * H * W))[:, :, None, None, None].expand(shape)
    report("mean_bwd", from_vol(dxa, N, C), want, **TOL[dtype])


@pytest.mark.parametrize("case", [(2, 1664, 1000), (12, 1000, 80), (1, 8, 1), (37, 300, 5)], ids=lambda c: "x".join(map(str, c)))
def test_linear_layers_of_the_classifier_tail(case):
    """nn.Linear forward / backward (models/densenet3d.py:139,158; models/resnt.py:34-35) on the pooled fp32 rows."""
    from segk import ops
    N, K, O = case
    x, w, b = rnd((N, K), torch.float32, 33), rnd((O, K), torch.float32, 34, K ** -0.5), rnd((O,), torch.float32, 35)
    dy = rnd((N, O), torch.float32, 36)
    y = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda())
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    ref = F.linear(xd, wd, bd)
    report("linear_fwd", y.cpu(), ref.float(), rtol=1e-5, atol=1e-5)
    ref.backward(dy.double())
    dw, db = torch.full((O, K), float("nan"), device="cuda"), torch.full((O,), float("nan"), device="cuda")
    dx = ops.linear_bwd(x.cuda(), w.cuda(), dy.cuda(), dw, db)
    report("linear_dx", dx.cpu(), xd.grad.float(), rtol=1e-5, atol=1e-5)
    report("linear_dw", dw.cpu(), wd.grad.float(), rtol=1e-5, atol=1e-5)
    report("linear_db", db.cpu(), bd.grad.float(), rtol=1e-5, atol=1e-5)
    ops.linear_bwd(x.cuda(), w.cuda(), dy.cuda(), dw, db, want_dx=False, accumulate=True)
    report("linear_dw_accumulated", dw.cpu(), 2 * wd.grad.float(), rtol=1e-5, atol=2e-5)
    report("linear_db_accumulated", db.cpu(), 2 * bd.grad.float(), rtol=1e-5, atol=2e-5)


def _run_detectmid(name, dtype, kw, shape):
    from models.resnt import DetectMid
    from oracle.ref_models import hash_init_, synthetic_volumes
    from segk import _lib
    n, depth, side = shape
    model = hash_init_(DetectMid(in_classes=1, **kw)).cuda().train().set_compute_dtype(dtype)
    x, t = synthetic_volumes(n, depth, side, tag=f"model:{name}")
    before = _lib.launches
    out = model(x.cuda())
    loss = F.mse_loss(out, t.cuda().float() / 80) if kw.get("if_regr") else F.cross_entropy(out, t.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert _lib.launches - before > 500, "the HIP library did not run the network"
    return model, x, out.detach(), loss.detach()


@pytest.mark.parametrize("name,kw,shape", [("detectmid", dict(), (2, 32, 128)), ("detectmid_regr", dict(if_regr=True, out_classes=1), (2, 32, 128))],
                         ids=["classify", "regress"])
def test_detectmid_fp32_vs_golden(name, kw, shape):
    """fp32 path against the reference's own outputs: class scores, loss, every parameter gradient's norm and leading
    values, BatchNorm running buffers, and the eval-mode output after the step."""
    g = np.load(GOLD)
    model, x, out, loss = _run_detectmid(name, torch.float32, kw, shape)
    want = torch.from_numpy(g[f"{name}/out"])
    assert rel_l2(out, want) < 1e-4, rel_l2(out, want)
    report(f"{name}/out", out.cpu(), want, rtol=1e-4, atol=1e-4 * float(want.abs().max()))
    report(f"{name}/loss", loss.cpu(), torch.from_numpy(g[f"{name}/loss"]), rtol=1e-4, atol=1e-6)
    if not kw.get("if_regr"):
        assert torch.equal(out.argmax(1).cpu(), want.argmax(1))
    # Gradients.  170 stacked layers whose BatchNorms see as few as 16 samples per channel: an activation within
    # rounding distance of a ReLU threshold flips its mask between two correct fp32 implementations, and everything
    # upstream of the flip moves by 1e-3..1e-1 (measured: the reference's own fp32 gradients sit 6e-3 (median) to 9e-2
    # away from an fp64 run of the same graph).  The HIP path is therefore held to that band against the fp64
    # oracle -- median over tensors no worse than twice the reference's, no tensor beyond 0.2 -- and the golden
    # norms from the reference guard against gross errors; the flip-free bound is test_small_densenet_fp32_grads.
    from oracle.ref_models import RefDetectMid, hash_init_, synthetic_volumes
    xs, ts = synthetic_volumes(*shape, tag=f"model:{name}")
    grads = {}
    for dt_ in (torch.float32, torch.float64):
        o = hash_init_(RefDetectMid(in_classes=1, **kw)).to(dt_).train()
        y = o(xs.to(dt_))
        (F.mse_loss(y, ts.to(dt_) / 80) if kw.get("if_regr") else F.cross_entropy(y, ts)).backward()
        grads[dt_] = {k: p.grad for k, p in o.named_parameters() if p.grad is not None}
    e_hips, e_refs = [], []
    for k, p in model.named_parameters():
        key = f"{name}/grad_norm/{k}"
        if key not in g.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        gn = float(g[key])
        assert abs(float(p.grad.double().norm()) - gn) <= 0.2 * gn + 1e-12, f"{k}: grad norm {float(p.grad.norm())} vs {gn}"
        e_hips.append(rel_l2(p.grad, grads[torch.float64][k]))
        e_refs.append(rel_l2(grads[torch.float32][k], grads[torch.float64][k]))
        assert e_hips[-1] < 0.2, f"{k}: rel-L2 to the fp64 oracle {e_hips[-1]:.3e} (reference fp32: {e_refs[-1]:.3e})"
    med_hip, med_ref = float(np.median(e_hips)), float(np.median(e_refs))
    print(f"{name}: gradient rel-L2 vs fp64 oracle: median hip {med_hip:.3e} / reference fp32 {med_ref:.3e}; max {max(e_hips):.3e} / {max(e_refs):.3e}")
    assert med_hip <= 2.0 * med_ref + 1e-4
    for k, b in model.named_buffers():
        if k.endswith("num_batches_tracked"):
            assert int(b) == 1, k
        else:
            want_sum = float(g[f"{name}/bufsum/{k}"])
            assert abs(float(b.double().sum()) - want_sum) <= 1e-4 * max(1.0, abs(want_sum)), k
    model.eval()
    with torch.no_grad():
        ev = model(x.cuda())
    we = torch.from_numpy(g[f"{name}/eval_out"])
    assert rel_l2(ev, we) < 1e-4, rel_l2(ev, we)


def test_small_densenet_fp32_grads():
    """A shallow DenseNet (two blocks of two layers, >= 64 samples per BatchNorm channel) has no activation at a ReLU
    threshold, so every parameter gradient must match the fp64 oracle as closely as the reference's fp32 run does."""
    from models.densenet3d import DenseNet
    from oracle.ref_models import RefDenseNet3D, hash_init_, hash_uniform
    cfg = dict(n_input_channels=3, conv1_t_size=3, block_config=(2, 2), num_init_features=64, num_classes=10)
    x = hash_uniform("small-densenet:x", (2, 3, 8, 32, 32), -1.7, 1.7)
    t = torch.tensor([3, 7])
    m = hash_init_(DenseNet(**cfg)).cuda().train().set_compute_dtype(torch.float32)
    out = m(x.cuda())
    F.cross_entropy(out, t.cuda()).backward()
    ref = {}
    for dt_ in (torch.float32, torch.float64):
        o = hash_init_(RefDenseNet3D(**cfg)).to(dt_).train()
        y = o(x.to(dt_))
        F.cross_entropy(y, t).backward()
        ref[dt_] = (y.detach(), {k: p.grad for k, p in o.named_parameters()})
    assert rel_l2(out, ref[torch.float64][0]) < 1e-5
    for k, p in m.named_parameters():
        e_hip = rel_l2(p.grad, ref[torch.float64][1][k])
        e_ref = rel_l2(ref[torch.float32][1][k], ref[torch.float64][1][k])
        assert e_hip <= 3.0 * e_ref + 2e-5, f"{k}: rel-L2 to the fp64 oracle {e_hip:.3e} (reference fp32: {e_ref:.3e})"


def test_detectmid_bf16():
    """bf16 path: class scores within the bf16 bound of the reference's fp32 scores relative to their spread; finite
    gradients for every parameter of the graph; deterministic run to run."""
    g = np.load(GOLD)
    outs = []
    for _ in range(2):
        model, x, out, loss = _run_detectmid("detectmid", torch.bfloat16, dict(), (2, 32, 128))
        outs.append((out.clone(), [None if p.grad is None else p.grad.clone() for p in model.parameters()]))
    want = torch.from_numpy(g["detectmid/out"])
    e = rel_l2(outs[0][0], want)
    print("detectmid bf16 rel-L2 of the class scores:", e, "loss", float(loss), "golden", float(g["detectmid/loss"]))
    assert e < 1e-1
    assert abs(float(loss) - float(g["detectmid/loss"])) < 5e-2 * float(g["detectmid/loss"])
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert (a is None and b is None) or (torch.isfinite(a).all() and torch.equal(a, b))


def test_detectmid_flat_grads_and_adam():
    """Flat gradient buffer + fused Adam on the volume network (the data-parallel wrapper of the 2-D networks)."""
    from models.resnt import DetectMid
    from oracle.ref_models import hash_init_, synthetic_volumes
    from segk.flat import FlatModel
    from segk.optim import Adam
    model = hash_init_(DetectMid(in_classes=1)).cuda().train().set_compute_dtype(torch.float32)
    ref = hash_init_(DetectMid(in_classes=1)).cuda().train().set_compute_dtype(torch.float32)
    x, t = synthetic_volumes(2, 16, 64, tag="model:flat")
    x, t = x.cuda(), t.cuda()
    F.cross_entropy(ref(x), t).backward()
    flat = FlatModel(model)
    opt = Adam(flat, lr=1e-4)
    for _ in range(2):          # second step: the analytically-zero gradients of the unused head are not rewritten
        loss = F.cross_entropy(model(x), t)
        loss.backward()
        if _ == 0:
            for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
                want = q.grad if q.grad is not None else torch.zeros_like(p)
                assert torch.equal(p.grad, want), k
        opt.step()
    assert torch.isfinite(flat.flat_param).all()


def test_detectmid_full_size_properties():
    """BASELINE volume shape (1 x 64 x 256 x 256, bf16), two volumes: size-independent properties instead of an oracle
    run -- bitwise run-to-run determinism of outputs and gradients, finite values, a gradient for every parameter of the
    graph, and (eval mode, running statistics) independence of a volume's scores from the rest of the batch."""
    from models.resnt import DetectMid
    from oracle.ref_models import hash_init_, synthetic_volumes
    model = hash_init_(DetectMid(in_classes=1)).cuda().train().set_compute_dtype(torch.bfloat16)
    x, t = synthetic_volumes(2, 64, 256, tag="bench:vol")
    x, t = x.cuda(), t.cuda()
    outs = []
    for _ in range(2):
        model.zero_grad()
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.reset_running_stats()
        out = model(x)
        F.cross_entropy(out, t).backward()
        outs.append((out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.isfinite(outs[0][0]).all()
    assert len(outs[0][1]) == sum(1 for k, _ in model.named_parameters() if not k.startswith("lastregr"))
    for k, g in outs[0][1].items():
        assert torch.isfinite(g).all() and torch.equal(g, outs[1][1][k]), k
    model.eval()
    with torch.no_grad():
        both = model(x)
        one = model(x[1:2])
    assert torch.equal(both[1:2], one)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [c for c in STEMS if c[7][0] == 1], ids=lambda c: f"{c[1]}to{c[5]}_k{c[6][0]}")
def test_stem_conv_factorised(dtype, case):
    """Strided Conv3d = per-slice (kh,kw) patch gather + depth-only (kd,1,1) convolution on the MFMA kernel: forward,
    patch gradient -> input gradient, and weight gradient against F.conv3d; volume faces and several volumes per batch."""
    from segk import ops
    N, Cin, D, H, W, Cout, k, s, p = case
    kd, kh, kw = k
    x = rnd((N, Cin, D, H, W), dtype, 1)
    w = rnd((Cout, Cin) + k, dtype, 2, scale=(2.0 / (Cin * kd * kh * kw)) ** 0.5)
    xa = to_vol(x, dtype)
    k2, s2, p2 = (1, kh, kw), (1, s[1], s[2]), (0, p[1], p[2])
    patches, _ = ops.im2col3d(xa, D, Cin, k2, s2, p2)
    Q, cs = Cin * kh * kw, (Cout + 7) // 8 * 8
    w1 = torch.zeros(cs, patches.C, kd)
    w1[:Cout, :Q] = w.permute(0, 3, 4, 1, 2).reshape(Cout, Q, kd)
    wf, wb = ops.convd_pack(w1.cuda(), dtype)
    y = ops.Act.empty(patches.N, patches.H, patches.W, cs, dtype, "cuda")
    part = ops.conv_stat_partials(y)
    ops.convd_fwd(patches, D, kd, wf, y, stat_partials=part)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    ref = F.conv3d(xr, wr, stride=s, padding=p)
    report("stem_fwd", from_vol(y, N, Cout), ref.detach().float(), **TOL[dtype])
    # the epilogue's partial sums give the batch statistics of the stored output
    want, got = torch.empty(2 * cs, device="cuda"), torch.empty(2 * cs, device="cuda")
    ops.norm_stats(y, 1, 1e-5, want)
    ops.norm_stats_from_partials(part, 1, y.P, 1e-5, got)
    report("stem_stats", got.cpu(), want.cpu(), rtol=2e-5, atol=2e-5)
    dy = rnd(tuple(ref.shape), dtype, 3)
    ref.backward(dy.double())
    dya = to_vol(dy, dtype)
    dpatches = ops.Act.empty(patches.N, patches.H, patches.W, patches.C, dtype, "cuda")
    ops.convd_fwd(dya, D, kd, wb, dpatches)
    dx = ops.Act.empty(N * D, H, W, 8, dtype, "cuda")
    ops.col2im3d(dpatches, dx, D, Cin, k2, s2, p2)
    tol = dict(TOL[dtype])
    if dtype == torch.bfloat16:
        tol["atol"] = 2e-2 * float(xr.grad.abs().max())
    report("stem_dgrad", from_vol(dx, N, Cin), xr.grad.float(), **tol)
    dw = torch.empty(cs, Q, kd, device="cuda")
    ops.convd_wgrad(patches, D, kd, dya, dw, cin_real=Q)
    got_w = dw[:Cout].view(Cout, kh, kw, Cin, kd).permute(0, 3, 4, 1, 2).cpu()
    assert rel_l2(got_w, wr.grad) < (1e-4 if dtype == torch.float32 else 2e-2)
    assert float(dw[Cout:].abs().max()) == 0.0 if cs > Cout else True
