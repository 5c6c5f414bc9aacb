"""GPU parity of every HIP kernel against the plain ATen op the reference calls (computed on the CPU in
fp32/fp64 from the same, dtype-rounded inputs).  All calls go through the C ABI (segk.ops -> ctypes).

Tolerances: fp32 path rel 1e-4 (the north-star bound; most kernels land at 1e-6), bf16 path rel 2e-2.
"""
import pytest
import torch
import torch.nn.functional as F

from util import report

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: dict(rtol=1e-4, atol=1e-4), torch.bfloat16: dict(rtol=2e-2, atol=2e-2)}


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).float()   # values exactly representable in dtype


def to_act(x, dtype, ld=None, c0=0):
    """NCHW fp32 (CPU) -> Act on the GPU, optionally as a channel slice of a wider buffer."""
    from segk.ops import Act
    N, C, H, W = x.shape
    ld = ld or C
    buf = torch.full((N, H, W, ld), 7.0, dtype=dtype, device="cuda")
    buf[..., c0:c0 + C] = x.permute(0, 2, 3, 1).to(dtype).cuda()
    return Act(buf, c0, C)


def from_act(a):
    return a.nchw().float().cpu()


# ---------------------------------------------------------------------------------------- convolution
CONV_CASES = [
    # N, H, W, Cin, Cout
    (2, 16, 16, 64, 64),
    (1, 24, 40, 128, 128),    # tiles not filling the image; BN=128 path
    (2, 12, 12, 96, 32),      # partial K-row (bf16), BN=32 path, image smaller than a tile
    (1, 32, 48, 32, 256),
    (3, 8, 16, 192, 96),      # 96-wide channel block
    (1, 16, 32, 32, 96),      # half K-row input (single k-substep); data gradient with 32 output channels
    (1, 16, 16, 128, 192),    # two 96-wide channel blocks
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_fwd_dgrad_wgrad(dtype, case):
    from segk import ops
    N, H, W, Cin, Cout = case
    x = rnd((N, Cin, H, W), dtype, 1)
    w = rnd((Cout, Cin, 3, 3), dtype, 2, scale=(2.0 / (9 * Cin)) ** 0.5)
    b = rnd((Cout,), torch.float32, 3)
    dy = rnd((N, Cout, H, W), dtype, 4)
    xa = to_act(x, dtype, ld=Cin + 16, c0=8)
    wf, wb = ops.conv_pack(w.cuda(), dtype)
    # forward (+bias, +ReLU)
    out = ops.Act.empty(N, H, W, Cout, dtype, "cuda")
    ops.conv_fwd(xa, wf, Cout, out, bias=b.cuda(), relu=True)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)).float()
    report("conv_fwd", from_act(out), ref, **TOL[dtype])
    # data gradient (+mask, into a slice), then accumulate a second time
    dya = to_act(dy, dtype)
    mask = rnd((N, Cin, H, W), dtype, 5)
    ma = to_act(mask, dtype)
    dx = ops.Act(torch.zeros(N, H, W, Cin + 8, dtype=dtype, device="cuda"), 8, Cin)
    ops.conv_fwd(dya, wb, Cin, dx, mask=ma)
    ref_dx = F.conv_transpose2d(dy.double(), w.double(), padding=1).float() * (mask > 0)
    report("conv_dgrad", from_act(dx), ref_dx, **TOL[dtype])
    ops.conv_fwd(dya, wb, Cin, dx, mask=ma, accumulate=True)
    t2 = dict(TOL[dtype]); t2["atol"] *= 2
    report("conv_dgrad_accum", from_act(dx), 2 * ref_dx, **t2)
    # mask on the leading channels only (a concat of a ReLU output and an unmasked skip tensor)
    mc = max(8, Cin // 2 // 8 * 8)
    ops.conv_fwd(dya, wb, Cin, dx, mask=ops.Act(ma.buf, 0, mc))
    ref_part = F.conv_transpose2d(dy.double(), w.double(), padding=1).float()
    ref_part[:, :mc] *= (mask[:, :mc] > 0)
    report("conv_dgrad_partial_mask", from_act(dx), ref_part, **TOL[dtype])
    # weight gradient
    dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
    ops.conv_wgrad(xa, dya, dw)
    ref_dw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=1).float()
    scale = float(ref_dw.abs().max())
    report("conv_wgrad", dw.cpu(), ref_dw, rtol=TOL[dtype]["rtol"], atol=TOL[dtype]["atol"] * max(scale, 1.0) * 0.05)


THIN_CASES = [
    # N, H, W, C0, C1, Cout: thin bf16 layers (one or two K-rows in) at sizes that fill the device several times over
    (2, 256, 256, 64, 0, 64),     # one K-row, 64-channel block (four workgroups per CU); same for its data gradient
    (2, 200, 328, 64, 0, 128),    # ragged borders, two channel blocks
    (2, 208, 320, 32, 64, 32),    # two segments (half K-row + K-row), 32-channel block; data gradient: 3 blocks of 32
    (2, 256, 272, 128, 0, 32),    # two K-rows
    (4, 128, 128, 128, 0, 256),   # more workgroups than fit the device at once (several resident rounds)
]


@pytest.mark.parametrize("case", THIN_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_thin_layers_bf16(case):
    """Thin layers (high-occupancy kernel variants with a single halo buffer): forward (+bias, ReLU), data gradient
    (+leading-channel mask, accumulate) against ATen in fp64."""
    from segk import ops
    dtype = torch.bfloat16
    N, H, W, C0, C1, Cout = case
    Cin = C0 + C1
    x = rnd((N, Cin, H, W), dtype, 51)
    w = rnd((Cout, Cin, 3, 3), dtype, 52, scale=(2.0 / (9 * Cin)) ** 0.5)
    b = rnd((Cout,), torch.float32, 53)
    dy = rnd((N, Cout, H, W), dtype, 54)
    wf, wb = ops.conv_pack(w.cuda(), dtype, seg0=C0 if C1 else 0)
    if C1:
        xin = (to_act(x[:, :C0], dtype, ld=C0 + 8, c0=8), to_act(x[:, C0:], dtype))
    else:
        xin = to_act(x, dtype, ld=Cin + 16, c0=8)
    out = ops.Act.empty(N, H, W, Cout, dtype, "cuda")
    ops.conv_fwd(xin, wf, Cout, out, bias=b.cuda(), relu=True)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)).float()
    report("thin_fwd", from_act(out), ref, **TOL[dtype])
    mask = rnd((N, Cin, H, W), dtype, 55)
    mc = max(8, Cin // 2 // 8 * 8)
    ma = to_act(mask[:, :mc], dtype)
    dya = to_act(dy, dtype, ld=Cout + 8)
    dx = ops.Act(torch.zeros(N, H, W, Cin + 8, dtype=dtype, device="cuda"), 8, Cin)
    ops.conv_fwd(dya, wb, Cin, dx, mask=ma)
    ref_dx = F.conv_transpose2d(dy.double(), w.double(), padding=1).float()
    ref_dx[:, :mc] *= (mask[:, :mc] > 0)
    report("thin_dgrad", from_act(dx), ref_dx, **TOL[dtype])
    ops.conv_fwd(dya, wb, Cin, dx, mask=ma, accumulate=True)
    t2 = dict(TOL[dtype]); t2["atol"] *= 2
    report("thin_dgrad_accum", from_act(dx), 2 * ref_dx, **t2)
    assert float(dx.buf[..., :8].float().abs().max()) == 0.0, "wrote outside the output slice"


@pytest.mark.parametrize("cin", [1, 3])
def test_first_conv_with_norm_recomputed(cin):
    """Conv2d(cin, 64) -> BatchNorm2d (batch statistics) -> ReLU, forward and backward, with the convolution's output
    recomputed from the image in every pass instead of stored (segk_conv_first_norm modes 1-4)."""
    from segk import ops
    from util import rel_l2
    N, H, W, C = 2, 20, 80, 64                      # partial tiles in both directions (tiles are 8 x 64 and 8 x 32 pixels)
    bf = torch.bfloat16
    x = rnd((N, cin, H, W), bf, 21)
    w = rnd((C, cin, 3, 3), bf, 22, scale=0.4)
    b = rnd((C,), torch.float32, 23, scale=0.3)
    gamma = 0.5 + torch.rand(C, generator=torch.Generator().manual_seed(24))
    beta = rnd((C,), torch.float32, 25, scale=0.5)
    g = rnd((N, C, H, W), bf, 26)
    eps, M = 1e-5, N * H * W
    # reference in double from the same bf16-representable operands
    z = F.conv2d(x.double(), w.double(), b.double(), padding=1).requires_grad_(True)
    gm, bt = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    mean, var = z.mean((0, 2, 3)), z.var((0, 2, 3), unbiased=False)
    zh = (z - mean[None, :, None, None]) / torch.sqrt(var + eps)[None, :, None, None]
    a_ref = torch.relu(zh * gm[None, :, None, None] + bt[None, :, None, None])
    (a_ref * g.double()).sum().backward()
    xc, wc, bc, gc, btc = x.cuda(), w.cuda(), b.cuda(), gamma.cuda(), beta.cuda()
    part = ops.conv_first_norm(1, xc, wc, bc)
    stats = torch.empty(2 * C, device="cuda")
    rmean, rvar = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    ops.norm_stats_from_partials(part, 1, M, eps, stats, rmean, rvar, 0.1)
    st = stats.view(C, 2).cpu().double()
    report("stem_mean", st[:, 0], mean.detach(), rtol=1e-4, atol=1e-5)
    report("stem_rstd", st[:, 1], 1.0 / torch.sqrt(var.detach() + eps), rtol=1e-4, atol=1e-5)
    report("stem_running_var", rvar.cpu(), 0.9 + 0.1 * var.detach() * M / (M - 1), rtol=1e-4, atol=1e-5)
    out = ops.Act.empty(N, H, W, C, bf, "cuda")
    ops.conv_first_norm(2, xc, wc, bc, stats=stats, gamma=gc, beta=btc, out=out)
    report("stem_out", from_act(out), a_ref.detach(), **TOL[bf])
    ga = to_act(g, bf, ld=C + 8)
    part2 = ops.conv_first_norm(3, xc, wc, bc, stats=stats, gamma=gc, beta=btc, g=ga)
    red = torch.empty(2 * C, device="cuda")
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.norm_bwd_sums_finalize(part2, red, dgamma=dg, dbeta=db)
    assert rel_l2(dg.cpu(), gm.grad) < 2e-3 and rel_l2(db.cpu(), bt.grad) < 2e-3, (rel_l2(dg.cpu(), gm.grad), rel_l2(db.cpu(), bt.grad))
    dz = ops.Act.empty(N, H, W, C, bf, "cuda")
    ops.conv_first_norm(4, xc, wc, bc, stats=stats, gamma=gc, beta=btc, red=red, g=ga, out=dz)
    assert rel_l2(from_act(dz), z.grad) < 1e-2, rel_l2(from_act(dz), z.grad)      # bf16 storage of dz: 4e-3
    report("stem_dz", from_act(dz), z.grad, rtol=2e-2, atol=2e-2 * float(z.grad.abs().max()))
    # the whole backward in two passes over g, dL/dz kept inside the weight-gradient kernel's LDS stage
    dw = torch.empty(C, cin, 3, 3, device="cuda")
    dg2, db2 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.conv_first_norm_bwd(xc, wc, bc, stats, gc, btc, ga, dw, dg2, db2)
    ref_dw = torch.nn.grad.conv2d_weight(x.double(), w.shape, z.grad, padding=1)
    assert rel_l2(dg2.cpu(), gm.grad) < 2e-3 and rel_l2(db2.cpu(), bt.grad) < 2e-3, (rel_l2(dg2.cpu(), gm.grad), rel_l2(db2.cpu(), bt.grad))
    assert rel_l2(dw.cpu(), ref_dw) < 1e-2, rel_l2(dw.cpu(), ref_dw)
    report("stem_dw", dw.cpu(), ref_dw, rtol=2e-2, atol=2e-2 * float(ref_dw.abs().max()))
    # ops.conv_first_norm_bwd issues the two halves as two calls (flags 4, then 8); one call with flags 0 gives the same bits
    from segk import _lib
    dw1 = torch.empty_like(dw)
    dg3, db3 = torch.empty_like(dg2), torch.empty_like(db2)
    ws = ops.workspace(_lib.query("segk_conv_first_norm_bwd_workspace", N, H, W), xc.device)
    _lib.call("segk_conv_first_norm_bwd", xc.data_ptr(), N, cin, H, W, wc.data_ptr(), bc.data_ptr(), stats.data_ptr(), gc.data_ptr(),
              btc.data_ptr(), ga.ptr, ga.ld, dw1.data_ptr(), dg3.data_ptr(), db3.data_ptr(), 0, ws.data_ptr(), ws.numel(), 0,
              torch.cuda.current_stream().cuda_stream)
    assert torch.equal(dw1, dw) and torch.equal(dg3, dg2) and torch.equal(db3, db2)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_conv_two_segments(dtype):
    from segk import ops
    N, H, W, C0, C1, Cout = 2, 16, 32, 96, 64, 64
    x0, x1 = rnd((N, C0, H, W), dtype, 11), rnd((N, C1, H, W), dtype, 12)
    w = rnd((Cout, C0 + C1, 3, 3), dtype, 13, scale=0.05)
    dy = rnd((N, Cout, H, W), dtype, 14)
    a0, a1 = to_act(x0, dtype), to_act(x1, dtype, ld=C1 + 32, c0=32)
    wf, wb = ops.conv_pack(w.cuda(), dtype, seg0=C0)
    out = ops.Act.empty(N, H, W, Cout, dtype, "cuda")
    ops.conv_fwd((a0, a1), wf, Cout, out)
    x = torch.cat([x0, x1], 1)
    report("conv_fwd_2seg", from_act(out), F.conv2d(x.double(), w.double(), padding=1).float(), **TOL[dtype])
    dya = to_act(dy, dtype)
    dx = ops.Act.empty(N, H, W, C0 + C1, dtype, "cuda")
    ops.conv_fwd(dya, wb, C0 + C1, dx)
    ref_dx = F.conv_transpose2d(dy.double(), w.double(), padding=1).float()
    report("conv_dgrad_2seg", from_act(dx), ref_dx, **TOL[dtype])
    # the same gradient written straight into the two segments' own buffers (second one accumulated)
    d0 = ops.Act(torch.zeros(N, H, W, C0 + 8, dtype=dtype, device="cuda"), 8, C0)
    d1 = ops.Act(torch.ones(N, H, W, C1, dtype=dtype, device="cuda"))
    ops.conv_fwd(dya, wb, C0 + C1, (d0, d1), accumulate1=True)
    report("conv_dgrad_2dst_a", from_act(d0), ref_dx[:, :C0], **TOL[dtype])
    report("conv_dgrad_2dst_b", from_act(d1), ref_dx[:, C0:] + 1.0, **TOL[dtype])
    assert float(d0.buf[..., :8].float().abs().max()) == 0.0
    # two destinations with the ReLU mask of the first segment's producer (a mask has to lie within the first destination)
    mask = rnd((N, C0, H, W), dtype, 15)
    m0, m1 = ops.Act.empty(N, H, W, C0, dtype, "cuda"), ops.Act.empty(N, H, W, C1, dtype, "cuda")
    ops.conv_fwd(dya, wb, C0 + C1, (m0, m1), mask=to_act(mask, dtype))
    report("conv_dgrad_2dst_masked_a", from_act(m0), ref_dx[:, :C0] * (mask > 0), **TOL[dtype])
    report("conv_dgrad_2dst_masked_b", from_act(m1), ref_dx[:, C0:], **TOL[dtype])
    with pytest.raises(Exception, match="within the first output"):
        ops.conv_fwd(dya, wb, C0 + C1, (m0, m1), mask=ops.Act.empty(N, H, W, C0 + 8, dtype, "cuda"))
    dw = torch.empty_like(w, device="cuda")
    ops.conv_wgrad((a0, a1), dya, dw)
    ref_dw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=1).float()
    report("conv_wgrad_2seg", dw.cpu(), ref_dw, rtol=TOL[dtype]["rtol"], atol=TOL[dtype]["atol"] * float(ref_dw.abs().max()) * 0.05)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, 12, 20, 64, 32), (1, 24, 24, 128, 64), (2, 6, 6, 256, 256)], ids=lambda c: "x".join(map(str, c)))
def test_conv_transpose_4s2p1(dtype, case):
    """nn.ConvTranspose2d(kernel 4, stride 2, padding 1) forward / data gradient / weight gradient."""
    from segk import ops
    N, H, W, Cin, Cout = case
    x = rnd((N, Cin, H, W), dtype, 21)
    w = rnd((Cin, Cout, 4, 4), dtype, 22, scale=(2.0 / (4 * Cin)) ** 0.5)
    b = rnd((Cout,), torch.float32, 23)
    dy = rnd((N, Cout, 2 * H, 2 * W), dtype, 24)
    xa = to_act(x, dtype, ld=Cin + 8, c0=8)
    wf, wb = ops.convt_pack(w.cuda(), dtype)
    out = ops.Act(torch.full((N, 2 * H, 2 * W, Cout + 16), 7.0, dtype=dtype, device="cuda"), 0, Cout)
    ops.convt_fwd(xa, wf, Cout, out, bias=b.cuda(), relu=True)
    ref = F.relu(F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=1)).float()
    report("convt_fwd", from_act(out), ref, **TOL[dtype])
    assert float(out.buf[..., Cout:].float().min()) == 7.0, "wrote outside the output slice"
    dya = to_act(dy, dtype, ld=Cout + 8)
    mask = rnd((N, Cin, H, W), dtype, 25)
    dx = ops.Act.empty(N, H, W, Cin, dtype, "cuda")
    ops.convt_dgrad(dya, wb, dx, mask=to_act(mask, dtype))
    ref_dx = F.conv2d(dy.double(), w.double(), stride=2, padding=1).float() * (mask > 0)
    report("convt_dgrad", from_act(dx), ref_dx, **TOL[dtype])
    ops.convt_dgrad(dya, wb, dx, mask=to_act(mask, dtype), accumulate=True)
    t2 = dict(TOL[dtype]); t2["atol"] *= 2
    report("convt_dgrad_accum", from_act(dx), 2 * ref_dx, **t2)
    dw = torch.empty(Cin, Cout, 4, 4, device="cuda")
    ops.convt_wgrad(xa, dya, dw)
    xr = x.double()
    wr = w.double().requires_grad_(True)
    F.conv_transpose2d(xr, wr, stride=2, padding=1).backward(dy.double())
    ref_dw = wr.grad.float()
    report("convt_wgrad", dw.cpu(), ref_dw, rtol=TOL[dtype]["rtol"], atol=TOL[dtype]["atol"] * max(float(ref_dw.abs().max()), 1.0) * 0.05)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_first_conv_im2col(dtype):
    from segk import ops
    N, H, W, Cin, Cout = 2, 24, 32, 3, 64
    x = rnd((N, Cin, H, W), torch.float32, 21)
    w = rnd((Cout, Cin, 3, 3), dtype, 22, scale=0.2)
    b = rnd((Cout,), torch.float32, 23)
    dy = rnd((N, Cout, H, W), dtype, 24)
    patches = ops.im2col3x3(x.cuda(), dtype)
    w1 = w.permute(0, 2, 3, 1).reshape(Cout, 27, 1, 1).contiguous().cuda()
    wf, _ = ops.conv_pack(w1, dtype, want_bwd=False)
    out = ops.Act.empty(N, H, W, Cout, dtype, "cuda")
    ops.conv_fwd(patches, wf, Cout, out, taps=1, bias=b.cuda())
    xr = x.to(dtype).float()
    report("first_conv", from_act(out), F.conv2d(xr.double(), w.double(), b.double(), padding=1).float(), **TOL[dtype])
    dw = torch.empty(Cout, 27, device="cuda")
    ops.conv_wgrad(patches, to_act(dy, dtype), dw, taps=1, cin_real=27)
    ref = torch.nn.grad.conv2d_weight(xr.double(), w.shape, dy.double(), padding=1).float()
    got = dw.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).cpu()
    report("first_conv_wgrad", got, ref, rtol=TOL[dtype]["rtol"], atol=TOL[dtype]["atol"] * float(ref.abs().max()) * 0.05)


@pytest.mark.parametrize("case", [(2, 24, 32, 3, 64), (1, 40, 72, 1, 64), (3, 9, 130, 3, 32), (1, 64, 64, 2, 64)],
                         ids=lambda c: "x".join(map(str, c)))
def test_first_conv_direct(case):
    """The first convolution straight from the fp32 image (bf16 output): values, ReLU flag, and the per-wave statistics
    partials (sums / sums of squares of what was stored).  Image sizes that do not fill the 8 x 64 workgroup tiles."""
    from segk import ops
    N, H, W, Cin, Cout = case
    x = rnd((N, Cin, H, W), torch.float32, 31)
    w = rnd((Cout, Cin, 3, 3), torch.bfloat16, 32, scale=0.2)
    b = rnd((Cout,), torch.float32, 33)
    assert ops.conv_first_direct_ok(x, Cout, torch.bfloat16) and not ops.conv_first_direct_ok(x, Cout, torch.float32)
    ref = F.conv2d(x.bfloat16().double(), w.double(), b.double(), padding=1).float()
    for relu in (False, True):
        buf = torch.full((N, H, W, Cout + 8), 7.0, dtype=torch.bfloat16, device="cuda")
        out = ops.Act(buf, 8, Cout)                      # a channel slice of a wider buffer
        part = ops.conv_first_fwd(x.cuda(), w.cuda(), b.cuda(), out, relu=relu, want_stats=True)
        want = ref.clamp_min(0) if relu else ref
        report("first_conv_direct", from_act(out), want, **TOL[torch.bfloat16])
        assert float(buf[..., :8].float().min()) == 7.0  # neighbouring channels untouched
        stored = from_act(out).double()
        sums = part.double().sum(0).cpu()                # [2, Cout]
        report("first_conv_direct/sum", sums[0], stored.sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
        report("first_conv_direct/sumsq", sums[1], (stored * stored).sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
        assert part.shape[0] % N == 0                    # rows are grouped per image (instance statistics)
        per = part.shape[0] // N
        report("first_conv_direct/sum[n=0]", part[:per].double().sum(0).cpu()[0], stored[0].sum((1, 2)), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("case", [(2, 24, 32, 3), (1, 40, 72, 1), (3, 9, 130, 3), (1, 64, 64, 2), (20, 64, 96, 3)],
                         ids=lambda c: "x".join(map(str, c)))
def test_first_conv_direct_wgrad(case):
    """Weight gradient of the first convolution straight from the image (many tiles per workgroup in the last case),
    against ATen on the bf16-rounded operands; accumulate flag; run-to-run bitwise reproducible."""
    from segk import ops
    N, H, W, Cin = case
    Cout = 64
    x = rnd((N, Cin, H, W), torch.float32, 41)
    dy = rnd((N, Cout, H, W), torch.bfloat16, 42)
    ref = torch.nn.grad.conv2d_weight(x.bfloat16().double(), (Cout, Cin, 3, 3), dy.double(), padding=1).float()
    dya = to_act(dy, torch.bfloat16, ld=Cout + 8, c0=8)
    dw = torch.full((Cout, Cin, 3, 3), 3.0, device="cuda")
    ops.conv_first_wgrad(x.cuda(), dya, dw)
    tol = dict(rtol=2e-2, atol=2e-2 * float(ref.abs().max()) * 0.05)
    report("first_conv_direct_wgrad", dw.cpu(), ref, **tol)
    again = torch.empty_like(dw)
    ops.conv_first_wgrad(x.cuda(), dya, again)
    assert torch.equal(dw, again)
    ops.conv_first_wgrad(x.cuda(), dya, again, accumulate=True)
    report("first_conv_direct_wgrad/accumulate", again.cpu(), 2 * ref, rtol=2e-2, atol=2 * tol["atol"])


# ---------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(4, 64, 24, 24), (2, 256, 8, 16), (3, 32, 48, 48), (2, 40, 12, 20)], ids=str)
def test_batchnorm_relu_fwd_bwd(dtype, shape):
    from segk import ops
    N, C, H, W = shape
    x = (rnd(shape, dtype, 31) * 1.5 + 0.3).to(dtype).float()
    gamma, beta = rnd((C,), torch.float32, 32) * 0.2 + 1.0, rnd((C,), torch.float32, 33) * 0.2
    dy = rnd(shape, dtype, 34)
    rm, rv = torch.zeros(C), torch.ones(C)
    xa = to_act(x, dtype, ld=C + 8, c0=8)
    stats = torch.empty(2 * C, device="cuda")
    rmg, rvg = rm.cuda(), rv.cuda()
    ops.norm_stats(xa, 1, 1e-5, stats, rmg, rvg, 0.1)
    out = ops.Act.empty(N, H, W, C, dtype, "cuda")
    ops.norm_apply(xa, 1, stats, gamma.cuda(), beta.cuda(), True, out)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rmd, rvd = rm.double(), rv.double()
    ref = F.relu(F.batch_norm(xd, rmd, rvd, gd, bd, True, 0.1, 1e-5))
    report("bn_relu_fwd", from_act(out), ref.float(), **TOL[dtype])
    report("running_mean", rmg.cpu(), rmd.float(), rtol=1e-4, atol=1e-5)
    report("running_var", rvg.cpu(), rvd.float(), rtol=1e-4, atol=1e-5)
    ref.backward(dy.double())
    dx = ops.Act.empty(N, H, W, C, dtype, "cuda")
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.norm_bwd(to_act(dy, dtype), xa, 1, stats, gamma.cuda(), beta.cuda(), True, dx, dgamma=dg, dbeta=db)
    # elements whose pre-activation is within rounding of zero may flip the ReLU mask in bf16: compare robustly
    report("bn_relu_bwd_dx", from_act(dx), xd.grad.float(), **TOL[dtype])
    n = N * H * W
    report("bn_dgamma", dg.cpu(), gd.grad.float(), rtol=TOL[dtype]["rtol"], atol=TOL[dtype]["atol"] * n ** 0.5)
    report("bn_dbeta", db.cpu(), bd.grad.float(), rtol=TOL[dtype]["rtol"], atol=TOL[dtype]["atol"] * n ** 0.5)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 64, 24, 16), (3, 40, 8, 12), (1, 256, 4, 6)], ids=str)
@pytest.mark.parametrize("skip_grad", [True, False], ids=["skip", "noskip"])
def test_batchnorm_relu_maxpool_fused(dtype, shape, skip_grad):
    """BatchNorm+ReLU with the 2x2 max pool behind it in one pass (forward and backward): against ATen in fp32,
    and against the separate norm / pool kernels (same rounding points) in both dtypes."""
    from segk import ops
    from util import rel_l2
    N, C, H, W = shape
    x = (rnd(shape, dtype, 41) * 1.5 + 0.3).to(dtype).float()
    gamma, beta = rnd((C,), torch.float32, 42) * 0.2 + 1.0, rnd((C,), torch.float32, 43) * 0.2
    dy = rnd(shape, dtype, 44)
    dp = rnd((N, C, H // 2, W // 2), dtype, 45)
    xa = to_act(x, dtype, ld=C + 8, c0=8)
    stats = torch.empty(2 * C, device="cuda")
    ops.norm_stats(xa, 1, 1e-5, stats)
    g, b = gamma.cuda(), beta.cuda()
    out = ops.Act(torch.zeros(N, H, W, C + 16, dtype=dtype, device="cuda"), 16, C)
    pool = ops.Act.empty(N, H // 2, W // 2, C, dtype, "cuda")
    ops.norm_pool_apply(xa, stats, g, b, out, pool)
    # separate kernels
    out2 = ops.Act.empty(N, H, W, C, dtype, "cuda")
    pool2 = ops.Act.empty(N, H // 2, W // 2, C, dtype, "cuda")
    ops.norm_apply(xa, 1, stats, g, b, True, out2)
    ops.maxpool_fwd(out2, pool2)
    assert torch.equal(from_act(out), from_act(out2)) and torch.equal(from_act(pool), from_act(pool2))
    dya, dpa = to_act(dy, dtype), to_act(dp, dtype)
    dx = ops.Act.empty(N, H, W, C, dtype, "cuda")
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.norm_pool_bwd(dya if skip_grad else None, dpa, xa, stats, g, b, dx, dgamma=dg, dbeta=db)
    da2 = to_act(dy if skip_grad else torch.zeros_like(dy), dtype)
    ops.maxpool_bwd(out2, dpa, da2, accumulate=True)
    dx2 = ops.Act.empty(N, H, W, C, dtype, "cuda")
    dg2, db2 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.norm_bwd(da2, xa, 1, stats, g, b, True, dx2, dgamma=dg2, dbeta=db2)
    bound = 1e-5 if dtype == torch.float32 else 1e-2     # bf16: the separate path rounds the summed gradient once more
    assert rel_l2(from_act(dx), from_act(dx2)) < bound
    assert rel_l2(dg, dg2) < bound and rel_l2(db, db2) < bound
    if dtype == torch.float32:
        xd = x.double().requires_grad_(True)
        gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
        a = F.relu(F.batch_norm(xd, None, None, gd, bd, True, 0.1, 1e-5))
        report("bn_pool_fwd", from_act(pool), F.max_pool2d(a, 2).float(), **TOL[dtype])
        loss = (F.max_pool2d(a, 2) * dp.double()).sum() + ((a * dy.double()).sum() if skip_grad else 0.0)
        loss.backward()
        report("bn_pool_bwd_dx", from_act(dx), xd.grad.float(), **TOL[dtype])
        report("bn_pool_dgamma", dg.cpu(), gd.grad.float(), rtol=1e-4, atol=1e-4 * (N * H * W) ** 0.5)
        report("bn_pool_dbeta", db.cpu(), bd.grad.float(), rtol=1e-4, atol=1e-4 * (N * H * W) ** 0.5)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("align", [False, True], ids=["ac0", "ac1"])
def test_instnorm_upsample_relu(dtype, align):
    from segk import ops
    N, C, H, W = 2, 64, 12, 20
    x = (rnd((N, C, H, W), dtype, 41) + 0.2).to(dtype).float()
    dy = rnd((N, C, 2 * H, 2 * W), dtype, 42)
    xa = to_act(x, dtype)
    stats = torch.empty(2 * N * C, device="cuda")
    ops.norm_stats(xa, N, 1e-5, stats)
    out = ops.Act(torch.zeros(N, 2 * H, 2 * W, C + 32, dtype=dtype, device="cuda"), 0, C)
    ops.upsample_fwd(xa, out, stats=stats, relu=True, align=align)
    xd = x.double().requires_grad_(True)
    ref = F.relu(F.interpolate(F.instance_norm(xd, eps=1e-5), scale_factor=2, mode="bilinear", align_corners=align))
    report("in_up_relu_fwd", from_act(out), ref.float(), **TOL[dtype])
    ref.backward(dy.double())
    dxh = ops.Act.empty(N, H, W, C, dtype, "cuda")
    ops.upsample_bwd(to_act(dy, dtype), out, dxh, relu=True, align=align)
    dx = ops.Act.empty(N, H, W, C, dtype, "cuda")
    ops.norm_bwd(dxh, xa, N, stats, None, None, False, dx)
    t = dict(TOL[dtype]); t["atol"] *= 4
    report("in_up_relu_bwd", from_act(dx), xd.grad.float(), **t)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_upsample_plain(dtype):
    from segk import ops
    N, C, H, W = 1, 32, 6, 10
    x, dy = rnd((N, C, H, W), dtype, 43), rnd((N, C, 2 * H, 2 * W), dtype, 44)
    for align in (False, True):
        out = ops.Act.empty(N, 2 * H, 2 * W, C, dtype, "cuda")
        ops.upsample_fwd(to_act(x, dtype), out, align=align)
        xd = x.double().requires_grad_(True)
        ref = F.interpolate(xd, scale_factor=2, mode="bilinear", align_corners=align)
        report(f"up_fwd_ac{align}", from_act(out), ref.float(), **TOL[dtype])
        ref.backward(dy.double())
        dx = ops.Act.empty(N, H, W, C, dtype, "cuda")
        ops.upsample_bwd(to_act(dy, dtype), None, dx, align=align)
        report(f"up_bwd_ac{align}", from_act(dx), xd.grad.float(), **TOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(1, 8, 1, 1), (2, 24, 3, 5), (1, 16, 2, 17), (1, 40, 33, 9), (1, 128, 5, 8), (2, 72, 9, 31)], ids=str)
def test_upsample_x2_edge_shapes(dtype, shape):
    """The block-wise x2 kernels on shapes that leave partial patches in every direction, one-pixel images and every
    channel grouping (1, 2, 4 and 8 vectors of 8 channels per block), both corner conventions, with and without the
    fused ReLU mask in the backward."""
    from segk import ops
    N, C, H, W = shape
    x, dy = rnd((N, C, H, W), dtype, 45), rnd((N, C, 2 * H, 2 * W), dtype, 46)
    for align in (False, True):
        for relu in (False, True):
            out = ops.Act(torch.zeros(N, 2 * H, 2 * W, C + 8, dtype=dtype, device="cuda"), 8, C)
            ops.upsample_fwd(to_act(x, dtype), out, relu=relu, align=align)
            xd = x.double().requires_grad_(True)
            ref = F.interpolate(xd, scale_factor=2, mode="bilinear", align_corners=align)
            got = from_act(out)
            report(f"up_fwd_ac{align}_relu{relu}", got, (F.relu(ref) if relu else ref).float(), **TOL[dtype])
            # the backward masks with the sign of the STORED output (a value within rounding of zero may have either sign
            # there), so the reference adjoint is taken with exactly that mask
            g = dy.double() * (got > 0).double() if relu else dy.double()
            ref.backward(g)
            dx = ops.Act.empty(N, H, W, C, dtype, "cuda")
            ops.upsample_bwd(to_act(dy, dtype), out if relu else None, dx, relu=relu, align=align)
            report(f"up_bwd_ac{align}_relu{relu}", from_act(dx), xd.grad.float(), **TOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_maxpool(dtype):
    from segk import ops
    N, C, H, W = 2, 64, 12, 16
    x = rnd((N, C, H, W), dtype, 51)
    x[:, :, ::4, ::4] = x[:, :, 1::4, 1::4]       # exact ties inside windows: first maximum must win
    dy = rnd((N, C, H // 2, W // 2), dtype, 52)
    xa = to_act(x, dtype, ld=C + 8)
    out = ops.Act.empty(N, H // 2, W // 2, C, dtype, "cuda")
    ops.maxpool_fwd(xa, out)
    xd = x.double().requires_grad_(True)
    ref = F.max_pool2d(xd, 2, 2)
    report("pool_fwd", from_act(out), ref.float(), rtol=0, atol=0)
    ref.backward(dy.double())
    dx = ops.Act(torch.ones(N, H, W, C, dtype=dtype, device="cuda"))
    ops.maxpool_bwd(xa, to_act(dy, dtype), dx, accumulate=True)
    report("pool_bwd_accum", from_act(dx), xd.grad.float() + 1, rtol=1e-6 if dtype == torch.float32 else 1e-2, atol=1e-6 if dtype == torch.float32 else 1e-2)
    ops.maxpool_bwd(xa, to_act(dy, dtype), dx)
    report("pool_bwd", from_act(dx), xd.grad.float(), rtol=0, atol=0)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_head_and_colsum(dtype):
    from segk import ops
    N, Cin, H, W, Cout = 2, 32, 16, 24, 4
    x = F.relu(rnd((N, Cin, H, W), dtype, 61))
    w, b = rnd((Cout, Cin), torch.float32, 62, 0.3), rnd((Cout,), torch.float32, 63)
    dl = rnd((N, Cout, H, W), torch.float32, 64)
    xa = to_act(x, dtype)
    logits = torch.empty(N, Cout, H, W, device="cuda")
    ops.head_fwd(xa, w.cuda(), b.cuda(), logits)
    xd = x.double().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv2d(xd, wd[:, :, None, None], bd)
    report("head_fwd", logits.cpu(), ref.float(), rtol=1e-5, atol=1e-5)
    ref.backward(dl.double())
    dx = ops.Act.empty(N, H, W, Cin, dtype, "cuda")
    dw, db = torch.empty(Cout, Cin, device="cuda"), torch.empty(Cout, device="cuda")
    ops.head_bwd(xa, w.cuda(), dl.cuda(), dx, True, dw, db)
    report("head_dx", from_act(dx), (xd.grad * (x > 0)).float(), **TOL[dtype])
    report("head_dw", dw.cpu(), wd.grad.float(), rtol=1e-4, atol=1e-3)
    report("head_db", db.cpu(), bd.grad.float(), rtol=1e-4, atol=1e-3)
    s = torch.empty(Cin, device="cuda")
    ops.colsum(xa, s)
    report("colsum", s.cpu(), x.double().sum((0, 2, 3)).float(), rtol=1e-5, atol=1e-3)
    ops.colsum(xa, s, accumulate=True)
    report("colsum_accum", s.cpu(), 2 * x.double().sum((0, 2, 3)).float(), rtol=1e-5, atol=2e-3)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(1, 8, 5, 7, 1), (2, 24, 9, 31, 3), (3, 64, 16, 17, 4), (1, 136, 6, 50, 8), (1, 1024, 3, 11, 8), (2, 64, 64, 80, 4)],
                         ids=lambda c: "x".join(map(str, c)))
def test_head_forward_row_staging(dtype, case):
    """1x1 head forward over the widths that select every staging mode (256, 128, 64 pixel rows per round in LDS, and rows too
    long for LDS), pixel counts that are not multiples of a round, 1..8 classes, input living in a channel slice."""
    from segk import ops
    N, Cin, H, W, Cout = case
    x = rnd((N, Cin, H, W), dtype, 65)
    w, b = rnd((Cout, Cin), torch.float32, 66, 0.3), rnd((Cout,), torch.float32, 67)
    xa = to_act(x, dtype, ld=Cin + 16, c0=8)
    logits = torch.full((N, Cout, H, W), float("nan"), device="cuda")
    ops.head_fwd(xa, w.cuda(), b.cuda(), logits)
    ref = F.conv2d(x.double(), w.double()[:, :, None, None], b.double())
    report("head_fwd", logits.cpu(), ref.float(), rtol=2e-5, atol=2e-5 * float(ref.abs().max()))


def test_copy_slice_and_adam():
    from segk import ops
    z = ops.Act(torch.full((2, 5, 7, 48), 3.0, dtype=torch.bfloat16, device="cuda"), 16, 24)
    ops.zero_slice(z)
    assert float(z.buf[..., 16:40].abs().max()) == 0 and float((z.buf[..., :16] - 3).abs().max()) == 0 and float((z.buf[..., 40:] - 3).abs().max()) == 0
    a = rnd((2, 40, 6, 6), torch.float32, 71)
    src = to_act(a, torch.float32)
    dst = ops.Act(torch.ones(2, 6, 6, 64, device="cuda"), 16, 40)
    ops.copy_slice(src, dst, accumulate=True)
    report("copy_accum", from_act(dst), a + 1, rtol=1e-6, atol=1e-6)
    assert float(dst.buf[..., :16].sub(1).abs().max()) == 0 and float(dst.buf[..., 56:].sub(1).abs().max()) == 0
    p0, g = rnd((1000,), torch.float32, 72), rnd((1000,), torch.float32, 73)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=1e-2)
    p, m, v = p0.cuda(), torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda")
    for step in (1, 2, 3):
        p_ref.grad = g * step
        opt.step()
        ops.adam_step(p, (g * step).cuda(), m, v, 1e-2, 0.9, 0.999, 1e-8, 0.0, step)
    report("adam", p.cpu(), p_ref.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("fine", [False, True], ids=["rows-per-wave", "four-rows-per-tile"])
@pytest.mark.parametrize("case", [(2, 16, 16, 64, 64, 9), (3, 20, 24, 128, 96, 9), (2, 12, 40, 32, 128, 9), (1, 9, 17, 64, 32, 9), (2, 16, 16, 32, 64, 1),
                                  (3, 24, 32, 64, 64, 9), (2, 40, 16, 32, 32, 9), (2, 16, 48, 96, 32, 9), (2, 8, 16, 32, 64, 9), (2, 16, 32, 64, 128, 9)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv_epilogue_statistics(dtype, case, fine):
    """Per-channel statistics produced by the convolution's epilogue (partial tiles, every channel-block width, 1x1
    taps, the thin layers of the weights-in-LDS kernel, both row layouts of the partial buffer) equal the statistics pass
    over the stored output, for batch (1 group) and instance (N groups) norms, including the running-buffer update."""
    from segk import ops
    N, H, W, Cin, Cout, taps = case
    k = 3 if taps == 9 else 1
    x = rnd((N, Cin, H, W), dtype, 1)
    w = rnd((Cout, Cin, k, k), dtype, 2, scale=(2.0 / (taps * Cin)) ** 0.5)
    b = rnd((Cout,), torch.float32, 3)
    xa = to_act(x, dtype)
    wf, _ = ops.conv_pack(w.cuda(), dtype, want_bwd=False)
    out = ops.Act(torch.zeros(N, H, W, Cout + 8, dtype=dtype, device="cuda"), 8, Cout)
    part = ops.conv_stat_partials(out, fine=fine)
    part.fill_(float("nan"))                                   # every row must be written
    ops.conv_fwd(xa, wf, Cout, out, taps=taps, bias=b.cuda(), stat_partials=part)
    assert torch.isfinite(part).all()
    for groups in (1, N):
        rm0, rv0 = torch.full((Cout,), 0.25, device="cuda"), torch.full((Cout,), 2.0, device="cuda")
        rm1, rv1 = rm0.clone(), rv0.clone()
        want = torch.empty(2 * groups * Cout, device="cuda")
        got = torch.empty_like(want)
        run = groups == 1
        ops.norm_stats(out, groups, 1e-5, want, rm0 if run else None, rv0 if run else None, 0.1)
        ops.norm_stats_from_partials(part, groups, N * H * W // groups, 1e-5, got, rm1 if run else None, rv1 if run else None, 0.1)
        report(f"stats groups={groups}", got.cpu(), want.cpu(), rtol=2e-5, atol=2e-5)
        if run:
            report("running_mean", rm1.cpu(), rm0.cpu(), rtol=1e-5, atol=1e-6)
            report("running_var", rv1.cpu(), rv0.cpu(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, 16, 16, 64, 64), (3, 20, 24, 96, 128), (2, 12, 40, 128, 32), (1, 9, 17, 32, 64), (2, 24, 24, 64, 96),
                                  (20, 64, 64, 64, 64)], ids=lambda c: "x".join(map(str, c)))
def test_conv_dgrad_with_norm_backward_sums(dtype, case):
    """Data gradient whose epilogue also reduces the two sums of the BatchNorm backward in front of it
    (conv -> BatchNorm -> ReLU -> conv, models/unetvgg.py:72-76; models/unetplusplus.py:20-25): the stored gradient is
    bitwise the plain data gradient, and dx / dgamma / dbeta equal those of the two-pass norm backward on it (every
    channel-block width, partial tiles, enough rows for the two-level reduction)."""
    from segk import ops
    N, H, W, Cup, C = case                                     # dy has Cup channels, the norm C
    dy = to_act(rnd((N, Cup, H, W), dtype, 1), dtype)
    w = rnd((Cup, C, 3, 3), dtype, 2, scale=(2.0 / (9 * C)) ** 0.5)
    _, wb = ops.conv_pack(w.cuda(), dtype, want_fwd=False)
    z = to_act(rnd((N, C, H, W), dtype, 3), dtype)
    gamma, beta = (rnd((C,), torch.float32, 4) * 0.3 + 1).cuda(), (rnd((C,), torch.float32, 5) * 0.3).cuda()
    stats = torch.empty(2 * C, device="cuda")
    ops.norm_stats(z, 1, 1e-5, stats)
    g0, g1 = ops.Act.empty(N, H, W, C, dtype, "cuda"), ops.Act(torch.zeros(N, H, W, C + 8, dtype=dtype, device="cuda"), 8, C)
    dx0, dx1 = ops.Act.empty(N, H, W, C, dtype, "cuda"), ops.Act.empty(N, H, W, C, dtype, "cuda")
    dg0, db0, dg1, db1 = (torch.empty(C, device="cuda") for _ in range(4))
    ops.conv_fwd(dy, wb, C, g0)
    ops.norm_bwd(g0, z, 1, stats, gamma, beta, True, dx0, dgamma=dg0, dbeta=db0)
    part = ops.conv_dgrad_norm(dy, wb, C, g1, z, stats, gamma, beta)
    assert torch.isfinite(part).all()
    assert torch.equal(g1.nchw(), g0.nchw())
    ops.norm_bwd_finish(part, g1, z, stats, gamma, beta, True, dx1, dgamma=dg1, dbeta=db1)
    tol = 2e-5 if dtype == torch.float32 else 1e-2             # bf16: dx is rounded after sums that differ in the last bits
    report("dgamma", dg1.cpu(), dg0.cpu(), rtol=2e-5, atol=2e-5 * float(dg0.abs().max()))
    report("dbeta", db1.cpu(), db0.cpu(), rtol=2e-5, atol=2e-5 * float(db0.abs().max()))
    report("dx", dx1.nchw().float().cpu(), dx0.nchw().float().cpu(), rtol=tol, atol=tol * float(dx0.nchw().float().abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_strided_2x2_convolution(dtype):
    """Conv2d(kernel_size=2, stride=2) (models/unetvgg.py:305) = 2x2 block gather + 1x1 convolution; forward, weight,
    bias and data gradients against torch, the data gradient also accumulating into a slice of a wider buffer."""
    from segk import ops
    from segk.engine import Runtime, Var
    N, C, Co, H, W = 2, 24, 16, 12, 20
    x, w, b = rnd((N, C, H, W), dtype, 81), rnd((Co, C, 2, 2), torch.float32, 82, 0.2), rnd((Co,), torch.float32, 83)
    dy = rnd((N, Co, H // 2, W // 2), dtype, 84)
    xa = ops.Act(torch.zeros(N, H, W, 64, device="cuda", dtype=dtype), 8, C)
    xa.nchw().copy_(x)
    wide = ops.block2(xa, ops.Act.empty(N, H // 2, W // 2, 4 * C, dtype, "cuda"))
    want = x.view(N, C, H // 2, 2, W // 2, 2).permute(0, 2, 4, 3, 5, 1).reshape(N, H // 2, W // 2, 4 * C)
    assert torch.equal(wide.buf.float().cpu(), want)                      # a permutation: bit-exact
    back = ops.Act(torch.ones(N, H, W, 64, device="cuda", dtype=dtype), 8, C)
    ops.block2(back, wide, backward=True, accumulate=True)
    assert torch.equal(from_act(back), (x + 1).to(dtype).float()) and float(back.buf[..., :8].sub(1).abs().max()) == 0
    # through the tape
    wp, bp = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    rt = Runtime({"w": wp, "b": bp}, {}, dtype, torch.device("cuda"), True, True)
    v = Var(xa)
    y = rt.conv_down2(v, "w", "b")
    xd, wd, bd = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xd, wd, bd, stride=2)
    ref.backward(dy)
    report("down2_y", from_act(y.act), ref.detach(), **TOL[dtype])
    rt.grad_of(y.act).nchw().copy_(dy)
    rt.backward_keep()
    report("down2_dx", from_act(rt.grad_of(xa)), xd.grad, **TOL[dtype])
    report("down2_dw", rt.grads["w"].cpu(), wd.grad, rtol=TOL[dtype]["rtol"], atol=1e-3 if dtype == torch.float32 else 5e-2)
    report("down2_db", rt.grads["b"].cpu(), bd.grad, rtol=1e-4, atol=1e-3)
    rt.release()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("scale", [8, 4, 0.5, 0.25])
def test_bilinear_resize_any_factor(dtype, scale):
    """nn.Upsample(scale_factor=s, mode='bilinear', align_corners=False) for s other than 2 (models/unetvgg.py:36 with
    upsc = 8, 4, 0.5): forward (also with fused InstanceNorm + ReLU) and the adjoint against torch."""
    from segk import ops
    N, C, H, W = 2, 16, 6 if scale >= 1 else 16, 10 if scale >= 1 else 24
    x = rnd((N, C, H, W), dtype, 91)
    Ho, Wo = int(H * scale), int(W * scale)
    dy = rnd((N, C, Ho, Wo), dtype, 92)
    xa = to_act(x, dtype, ld=32, c0=8)
    xd = x.clone().requires_grad_(True)
    ref = torch.nn.functional.interpolate(xd, scale_factor=scale, mode="bilinear", align_corners=False)
    ref.backward(dy)
    out = ops.resize_fwd(xa, ops.Act.empty(N, Ho, Wo, C, dtype, "cuda"), scale)
    report(f"resize_x{scale}", from_act(out), ref.detach(), **TOL[dtype])
    dx = ops.Act(torch.ones(N, H, W, 32, device="cuda", dtype=dtype), 8, C)
    ops.resize_bwd(to_act(dy, dtype), dx, scale)
    report(f"resize_x{scale}_dx", from_act(dx), xd.grad, **TOL[dtype])
    ops.resize_bwd(to_act(dy, dtype), dx, scale, accumulate=True)
    report(f"resize_x{scale}_dx_accum", from_act(dx), 2 * xd.grad, rtol=2 * TOL[dtype]["rtol"], atol=2 * TOL[dtype]["atol"])
    assert float(dx.buf[..., :8].sub(1).abs().max()) == 0
    # fused InstanceNorm (statistics of the low-resolution input) + ReLU
    stats = torch.empty(2 * N * C, device="cuda")
    ops.norm_stats(xa, N, 1e-5, stats)
    fused = ops.resize_fwd(xa, ops.Act.empty(N, Ho, Wo, C, dtype, "cuda"), scale, stats=stats, relu=True)
    want = torch.relu(torch.nn.functional.interpolate(torch.nn.functional.instance_norm(x), scale_factor=scale, mode="bilinear", align_corners=False))
    report(f"in_resize_x{scale}_relu", from_act(fused), want, **TOL[dtype])
