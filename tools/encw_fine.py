"""Dev tool (GPU box): phase stamps of the wave-per-32-cells encode kernel (build: tools/build_variant.py encwfine codebook_encode_wave.hip -DQV2X_ENCW_FINE).
    python tools/encw_fine.py <tag> <n_frames>      s_memtime ticks of the first 4096 waves"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quantv2x_amd import lib as L
tag, n = sys.argv[1], int(sys.argv[2])
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_{tag}.so")
from quantv2x_amd.engine import deploy
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
h, w = 100, 352
x = torch.randint(-128, 127, (n, h + 2, w + 2, 256), dtype=torch.int8, device="cuda")
codes = torch.empty((3, n * h * w), dtype=torch.uint8, device="cuda")
d = L.EncodeDesc(); d.n, d.h, d.w, d.levels, d.kc = n, h, w, 3, 128; d.in_zx, d.in_delta = 0, 0.05
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    e0.record()
    L.check(eng.lib.qv2x_codebook_encode_wave_f32(C.byref(d), L.ptr(x), None, eng.level_ptrs, L.ptr(codes), L.current_stream()))
    e1.record()
torch.cuda.synchronize()
raw = C.CDLL(L.LIB_PATH)
nb = 4096
buf = np.zeros((nb, 32), np.int64)
assert raw.qv2x_debug_encw_fine(buf.ctypes.data_as(C.c_void_p), nb) == 0
names = ["stage GEMM", "read back z", "qhead GEMM", "read back q + |q|^2", "distance GEMM + argmin", "lhead GEMM (+ gathers)", "read back x"]
tot = buf[:, 21] - buf[:, 0]
print(f"{nb} waves of a {n}-frame launch ({e0.elapsed_time(e1) * 1e3:.0f} us): whole wave (level 0 start .. level 2 argmin) mean {tot.mean():.0f} ticks, min {tot.min()}, max {tot.max()}")
for l in range(3):
    b = buf[:, 8 * l:8 * l + 8]
    k = 8 if l < 2 else 6
    dd = np.diff(b[:, :k], axis=1)
    print(f"level {l}: " + ", ".join(f"{names[i]} {dd[:, i].mean():.0f}" for i in range(k - 1)))
f2 = np.diff(buf[:, 24:31], axis=1)
print("level 1, stage pair 1: acc init, groups 0-7, 8-15, 16-23, 24-31 (64 MFMAs = 4096 cycles each), tile stores:", [int(v) for v in f2.mean(axis=0)])
print("ideal: 256 MFMAs x 64 cycles = 16384 cycles per tile pair -- 65536 per 256 x 256 GEMM, 32768 for the 128 codes")
