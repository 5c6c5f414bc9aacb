"""In-kernel clock of the convolution main loop: run `make -C image-segmentation_amd/csrc probe`, then
    SEGK_LIB_PATH=image-segmentation_amd/libsegk_probe.so python scratch/clock_probe.py
Launches one layer back to back for ~2.5 s on random data, then reads the shader-clock / real-time stamps of the last launch."""
import ctypes, os, sys, time, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-segmentation_amd"))
from segk import ops, _lib
from segk.ops import Act
N = 32
for name, H, Cin, Cout in [("conv3.17", 96, 256, 256), ("conv4.27", 48, 512, 512), ("dec3.0", 96, 512, 256), ("conv2.10", 192, 128, 128), ("conv1.3", 384, 64, 64), ("dec2.2", 192, 128, 32), ("dec3.2", 96, 256, 64)]:
    x = Act(torch.randn(N, H, H, Cin, device="cuda").to(torch.bfloat16))
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    wf, wb = ops.conv_pack(w, torch.bfloat16)
    y = Act.empty(N, H, H, Cout, torch.bfloat16, "cuda")
    fl = 2.0 * N * H * H * Cin * Cout * 9
    t0 = time.time(); n = 0
    while time.time() - t0 < 2.5:
        for _ in range(50): ops.conv_fwd(x, wf, Cout, y)
        torch.cuda.synchronize(); n += 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv_fwd(x, wf, Cout, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    buf = (ctypes.c_ulonglong * 4096)()
    h = _lib.lib()
    h.segk_debug_clock_probe.argtypes = [ctypes.c_void_p]
    assert h.segk_debug_clock_probe(buf) == 0
    ghz = [buf[4 * i] / buf[4 * i + 1] * 0.1 for i in range(1024) if buf[4 * i + 1] > 0]
    cyc = [buf[4 * i] for i in range(1024) if buf[4 * i + 1] > 0]
    clk = statistics.median(ghz)
    pro = statistics.median([buf[4 * i + 2] for i in range(1024) if buf[4 * i + 1] > 0])
    tot = statistics.median([buf[4 * i + 3] for i in range(1024) if buf[4 * i + 1] > 0])
    nmfma = fl / 16384.0
    busy = nmfma * 16 / 1024 / (ms * 1e-3 * clk * 1e9)
    print(f"{name}: {ms*1e3:.1f} us {fl/ms/1e9:.0f} TF | in-kernel clock median {clk:.3f} GHz (min {min(ghz):.3f} max {max(ghz):.3f}) | "
          f"workgroup lifetime {tot:.0f} cycles = prologue (first halo + weights landed) {pro:.0f} + main loop {statistics.median(cyc):.0f} + epilogue {tot - pro - statistics.median(cyc):.0f} | MFMA pipe busy at that clock {100*busy:.0f} %", flush=True)
