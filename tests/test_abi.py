"""The C-ABI library loads and exports every symbol include/segk.h declares, with the ctypes prototypes the
Python boundary uses (no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, "include", "segk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.findall(r"(?:int64_t|int|size_t|const char\*)\s+(segk_\w+)\s*\(([^)]*)\)\s*;", src)


def test_header_matches_ctypes_prototypes():
    from segk import _lib
    kinds = {"int": _lib.I, "int64_t": _lib.L, "float": _lib.F, "size_t": _lib.Z}
    decls = _header_decls()
    assert len(decls) >= 30
    for name, args in decls:
        want = []
        for a in args.split(","):
            a = a.strip()
            if a == "void":
                continue
            want.append(_lib.P if "*" in a else kinds[" ".join(a.split()[:-1])])
        assert _lib._PROTOS[name][1] == want, name
    assert set(_lib._PROTOS) == {n for n, _ in decls}


def test_library_exports_every_declared_symbol():
    from segk import _lib
    if not _lib.available():
        pytest.fail(f"{_lib.LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    h = ctypes.CDLL(_lib.LIB_PATH)
    for name, _ in _header_decls():
        assert hasattr(h, name), name
    assert _lib.query("segk_version") >= 100
    assert _lib.query("segk_conv_krow", _lib.SEGK_BF16) == 64 and _lib.query("segk_conv_krow", _lib.SEGK_F32) == 32
    # pure host-side queries are safe without a GPU
    assert _lib.query("segk_dice_focal_workspace", 32, 4, 384 * 384) > 0
    assert _lib.query("segk_conv_wgrad_workspace", _lib.SEGK_BF16, 9, 64, 0, 32, 384, 384, 64) > 0
    assert _lib.query("segk_conv_packed_bytes", _lib.SEGK_BF16, 64, 96, 9, 0) == 2 * 9 * 64 * 128


def test_argument_validation_needs_no_gpu():
    """Bad arguments are rejected on the host before any HIP call, with a readable message."""
    from segk import _lib
    with pytest.raises(_lib.SegkError, match="SEGK_EINVAL"):
        _lib.call("segk_conv_pack", None, 64, 64, 9, 0, _lib.SEGK_BF16, None, None, 0, None)
    with pytest.raises(_lib.SegkError, match="taps"):
        _lib.call("segk_conv_pack", 16, 64, 64, 4, 0, _lib.SEGK_BF16, 16, None, 0, None)
    with pytest.raises(_lib.SegkError, match="SEGK_ESHAPE"):
        _lib.call("segk_maxpool2x2_fwd", _lib.SEGK_F32, 16, 8, 1, 3, 4, 8, 16, 8, 0, None)
    with pytest.raises(_lib.SegkError, match="aligned"):
        _lib.call("segk_norm_apply", _lib.SEGK_F32, 4, 8, 1, 16, 8, 16, None, None, 0, 16, 8, 0, None)
    # entry points added for the first convolution, the norm-backward sums, the static-scratch loss and the classifier tail
    F32, BF16 = _lib.SEGK_F32, _lib.SEGK_BF16
    with pytest.raises(_lib.SegkError, match="SEGK_E"):       # Cin of the direct first convolution is 1..3
        _lib.call("segk_conv_first_fwd", 16, 1, 4, 16, 16, 16, None, 64, 16, BF16, 64, None, 0, 0, None)
    with pytest.raises(_lib.SegkError, match="SEGK_E"):       # ... and it writes bf16 only
        _lib.call("segk_conv_first_fwd", 16, 1, 3, 16, 16, 16, None, 64, 16, F32, 64, None, 0, 0, None)
    with pytest.raises(_lib.SegkError, match="conv_dgrad_norm"):   # z, stats and partials are mandatory
        _lib.call("segk_conv_dgrad_norm", BF16, 9, 16, 64, 64, 1, 16, 16, 16, 64, 16, 64, None, 64, None, None, None, None, 0, None)
    with pytest.raises(_lib.SegkError, match="frozen"):
        _lib.call("segk_norm_bwd_finish", F32, 16, 4, 16, 8, 16, 8, 64, 8, 16, None, None, 4, 16, 8, None, None, 16, 1 << 20, 0, None)
    with pytest.raises(_lib.SegkError, match="SEGK_EWORKSPACE"):
        _lib.call("segk_norm_bwd_finish", F32, 16, 4096, 16, 8, 16, 8, 64, 8, 16, None, None, 1, 16, 8, None, None, 16, 16, 0, None)
    with pytest.raises(_lib.SegkError, match="static scratch"):    # more block partials than the 1 MiB array holds
        _lib.call("segk_dice_focal_fwd_static", 16, F32, 16, None, 65535, 8, 4096, 2.0, None, 0, 16, 16, None, 0, None)
    with pytest.raises(_lib.SegkError, match="SEGK_ESHAPE"):
        _lib.call("segk_linear_fwd", 16, 16, None, 4, 0, 8, 16, 0, None)
    with pytest.raises(_lib.SegkError, match="SEGK_EINVAL"):
        _lib.call("segk_linear_bwd", 16, 16, None, 4, 8, 8, None, 16, None, 0, 0, None)
    with pytest.raises(_lib.SegkError, match="SEGK_E"):
        _lib.call("segk_rows_mean_fwd", BF16, 16, 12, 2, 4, 12, 16, 0, None)     # 12 channels: not a multiple of 8
    # the forward convolution packs (pixel index << 3) into 32 bits: 2^28 pixels and more per launch are refused, not mis-addressed
    with pytest.raises(_lib.SegkError, match="2\\^28"):
        _lib.call("segk_conv_fwd", BF16, 9, 16, 64, 64, None, 0, 0, 1 << 12, 256, 256, 16, 64, None, 16, 64, None, 0, 0, None, 0, 0, None, 0, 0, None)
    # the fused first layer (convolution output recomputed): modes 1..4, 64 output channels, statistics for modes 2-4
    with pytest.raises(_lib.SegkError, match="mode"):
        _lib.call("segk_conv_first_norm", 5, 16, 1, 3, 16, 16, 16, None, 64, None, None, None, None, None, 0, None, 0, 16, 0, None)
    with pytest.raises(_lib.SegkError, match="statistics"):
        _lib.call("segk_conv_first_norm", 2, 16, 1, 3, 16, 16, 16, None, 64, None, None, None, None, None, 0, 16, 64, None, 0, None)
    with pytest.raises(_lib.SegkError, match="SEGK_EWORKSPACE"):
        _lib.call("segk_conv_first_norm_bwd", 16, 1, 3, 16, 16, 16, None, 16, None, None, 16, 64, 16, None, None, 0, 16, 16, 0, None)
    # a bias that is only 4-byte aligned is refused by the three first-layer entry points (their kernels read it 16 bytes at a time)
    with pytest.raises(_lib.SegkError, match="bias must be 16-byte aligned"):
        _lib.call("segk_conv_first_fwd", 16, 1, 3, 16, 16, 16, 20, 64, 16, BF16, 64, None, 0, 0, None)
    with pytest.raises(_lib.SegkError, match="bias must be 16-byte aligned"):
        _lib.call("segk_conv_first_norm", 1, 16, 1, 3, 16, 16, 16, 20, 64, None, None, None, None, None, 0, None, 0, 16, 0, None)
    with pytest.raises(_lib.SegkError, match="bias must be 16-byte aligned"):
        _lib.call("segk_conv_first_norm_bwd", 16, 1, 3, 16, 16, 16, 20, 16, None, None, 16, 64, 16, None, None, 0, 16, 1 << 30, 0, None)
    with pytest.raises(_lib.SegkError, match="pairs"):
        _lib.call("segk_norm_bwd_sums_finalize", 16, 4, 64, 16, 16, None, 0, None, 0, 0, None)
    assert _lib.query("segk_conv_first_norm_bwd_workspace", 32, 384, 384) > 512 * 64 * 32 * 4
    assert _lib.query("segk_norm_bwd_finish_workspace", 4096, 64) >= 2 * 64 * 4
    assert _lib.query("segk_conv_first_stat_rows", 32, 384, 384) > 0

