"""The loss forward that allocates nothing (`segk_dice_focal_fwd_static`: block partials in a fixed device array of the
library) against `segk_dice_focal_fwd` (caller-provided workspace), through the C ABI: bit-identical per-(image, class)
sums and foreground flags on two calls in a row, the second with a smaller input, and a clear error beyond the array."""
import ctypes

import pytest
import torch

from segk import _lib

pytestmark = pytest.mark.gpu

def _static_fn():
    return getattr(_lib.lib(), "segk_dice_focal_fwd_static")


def _both(N, C, side, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    z = (torch.randn(N, C, side, side, generator=g) * 3).to(dtype).cuda()
    t = torch.randint(0, C, (N, side, side), generator=g)
    t[N // 2] = 0                                       # one image without foreground
    t = t.cuda()
    HW = side * side
    dev = torch.cuda.current_device()
    st = torch.cuda.current_stream().cuda_stream
    code = _lib.SEGK_BF16 if dtype == torch.bfloat16 else _lib.SEGK_F32
    out = []
    for static in (False, True):
        sums = torch.full((N * C * 4,), float("nan"), device="cuda")
        flags = torch.full((N,), -7, dtype=torch.int32, device="cuda")
        if static:
            rc = _static_fn()(z.data_ptr(), code, t.data_ptr(), None, N, C, HW, 2.0, None, 0, sums.data_ptr(), flags.data_ptr(),
                              None, dev, st)
            assert rc == 0, _lib.lib().segk_last_error().decode()
        else:
            nbytes = _lib.query("segk_dice_focal_workspace", N, C, HW)
            ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            _lib.call("segk_dice_focal_fwd", z.data_ptr(), code, t.data_ptr(), None, N, C, HW, 2.0, None, 0, sums.data_ptr(),
                      flags.data_ptr(), None, ws.data_ptr(), nbytes, dev, st)
        torch.cuda.synchronize()
        out.append((sums.cpu(), flags.cpu()))
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_static_scratch_forward_matches_workspace_forward(dtype):
    # two calls in a row on the same device array: the second, smaller one must not see the first one's partials
    for N, C, side, seed in ((6, 4, 192, 1), (3, 3, 40, 2)):
        (s0, f0), (s1, f1) = _both(N, C, side, dtype, seed)
        assert torch.isfinite(s0).all() and torch.equal(s0, s1)          # bit for bit
        assert torch.equal(f0, f1) and set(f0.tolist()) <= {0, 1} and f0[N // 2] == 0


def test_static_scratch_forward_inside_a_captured_graph():
    """After one warm-up call on the device (the symbol look-up) the entry point only launches kernels: captured into a
    HIP graph and replayed, it gives the workspace path's sums bit for bit."""
    N, C, side = 4, 4, 96
    (s_ref, f_ref), _ = _both(N, C, side, torch.float32, 5)              # also the warm-up call of the static entry point
    g = torch.Generator().manual_seed(5)
    z = (torch.randn(N, C, side, side, generator=g) * 3).cuda()
    t = torch.randint(0, C, (N, side, side), generator=g)
    t[N // 2] = 0
    t = t.cuda()
    sums = torch.zeros(N * C * 4, device="cuda")
    flags = torch.zeros(N, dtype=torch.int32, device="cuda")
    dev = torch.cuda.current_device()
    graph = torch.cuda.CUDAGraph()
    side_stream = torch.cuda.Stream()
    side_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side_stream):
        with torch.cuda.graph(graph, stream=side_stream):
            rc = _static_fn()(z.data_ptr(), _lib.SEGK_F32, t.data_ptr(), None, N, C, side * side, 2.0, None, 0, sums.data_ptr(),
                              flags.data_ptr(), None, dev, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _lib.lib().segk_last_error().decode()
    for _ in range(2):
        sums.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(sums.cpu(), s_ref) and torch.equal(flags.cpu(), f_ref)


def test_static_scratch_rejects_oversized_shapes():
    # 1 MiB of partials: 65535 images of 8 classes need more; the check is on the host, before any launch
    rc = _static_fn()(16, _lib.SEGK_F32, 16, None, 65535, 8, 4096, 2.0, None, 0, 16, 16, None, 0, None)
    assert rc == -4                                                      # SEGK_EWORKSPACE
    assert b"static scratch" in _lib.lib().segk_last_error()
    assert isinstance(ctypes.c_int(rc).value, int)
