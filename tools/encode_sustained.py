"""Dev tool (GPU box): the 32-frame encode launched back to back -- per-launch device time as the run goes on (clock / power behaviour under
sustained f32 MFMA load).      python tools/encode_sustained.py [frames] [launches]"""
import ctypes as C, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantv2x_amd import lib as L
if os.environ.get("QV2X_LIB_TAG"):
    L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_{os.environ['QV2X_LIB_TAG']}.so")
from quantv2x_amd.engine import deploy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
h, w = 100, 352
x = torch.randint(-128, 127, (n, h + 2, w + 2, 256), dtype=torch.int8, device="cuda")
codes = torch.empty((3, n * h * w), dtype=torch.uint8, device="cuda")
d = L.EncodeDesc(); d.n, d.h, d.w, d.levels, d.kc = n, h, w, 3, 128; d.in_zx, d.in_delta = 0, 0.05
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
torch.cuda.synchronize()
ev[0].record()
for i in range(K):
    L.check(eng.lib.qv2x_codebook_encode_f32(C.byref(d), L.ptr(x), eng.level_ptrs, L.ptr(codes), L.current_stream()))
    ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(K)]
print("per-launch us:", " ".join(f"{i}:{t[i]:.0f}" for i in (0, 1, 2, 4, 9, 19, 49, 99, 149, K - 1) if i < K))
print(f"mean of the last half: {sum(t[K // 2:]) / (K - K // 2):.0f} us = {2 * 21.92 * n / (sum(t[K // 2:]) / (K - K // 2)) * 1e3:.1f} TFLOP/s")
try:
    print(subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout[-1500:])
except Exception as e:
    print("rocm-smi:", e)
