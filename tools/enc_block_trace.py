"""Where and when every encode workgroup ran (dev tool, runs on the GPU box): builds libqv2x with -DQV2X_ENC_TRACE."""
import os, sys, subprocess, glob, ctypes as C, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantv2x_amd import lib as L
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L.LIB_PATH = os.path.join(ROOT, "tools", "cache", "libqv2x_trace.so")
if not os.path.exists(L.LIB_PATH):
    from quantv2x_amd import build as B
    subprocess.check_call(["hipcc"] + B.FLAGS + ["-DQV2X_ENC_TRACE", "-o", L.LIB_PATH] + B.SOURCES)
from quantv2x_amd.engine import deploy
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
h, w = 100, 352
x = torch.randint(-128, 127, (1, h + 2, w + 2, 256), dtype=torch.int8, device="cuda")
codes = torch.empty((3, h * w), dtype=torch.uint8, device="cuda")
d = L.EncodeDesc(); d.n, d.h, d.w, d.levels, d.kc = 1, h, w, 3, 128; d.in_zx, d.in_delta = 0, 0.05
for _ in range(3):
    L.check(eng.lib.qv2x_codebook_encode_f32(C.byref(d), L.ptr(x), eng.level_ptrs, L.ptr(codes), L.current_stream()))
torch.cuda.synchronize()
n = 1100
buf = (C.c_ulonglong * (4 * n))()
eng.lib.qv2x_debug_encode_blocks(buf, n)
rows = [tuple(buf[4 * i + k] for k in range(4)) for i in range(n)]
t0 = min(r[0] for r in rows)
per_cu = collections.defaultdict(list)
for i, (s, e, hw, xcc) in enumerate(rows):
    cu = (xcc & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)
    per_cu[cu].append((s - t0, e - t0, i))
print("CUs used:", len(per_cu), " blocks per CU: min", min(len(v) for v in per_cu.values()), "max", max(len(v) for v in per_cu.values()))
dur = sorted(e - s for s, e, _, _ in rows)
print("block duration ticks: min", dur[0], "median", dur[len(dur) // 2], "max", dur[-1])   # counters of different XCDs are not aligned: compare within a CU only
# concurrency per CU
conc = collections.Counter()
for cu, lst in per_cu.items():
    ev = sorted([(s, 1) for s, e, _ in lst] + [(e, -1) for s, e, _ in lst])
    c = m = 0
    for _, dlt in ev:
        c += dlt; m = max(m, c)
    conc[m] += 1
print("max concurrent blocks per CU -> number of CUs:", dict(conc))
cu0 = sorted(per_cu.items())[0]
print("one CU", cu0[0], [(s, e, i) for s, e, i in sorted(cu0[1])])
