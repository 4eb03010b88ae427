"""Dev tool: the two-stage exact encode against the exact encode at the bench's batch (32 V2X-Real frames of golden-like rows), HIP events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import bench
from quantv2x_amd import lib as L
if os.environ.get("QV2X_LIB_TAG"):
    L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_{os.environ['QV2X_LIB_TAG']}.so")

def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    state, eng, _, _ = bench.build_engine(min(32, os.cpu_count() or 8))
    _, full, _, _ = bench.frame_batch(1, 0, frames, torch.device("cuda"))
    eng.encode_mode = "exact"
    eng(full)                                              # fills the workspace's shrinker output with real rows
    torch.cuda.synchronize()
    exact = eng.encode_codes(frames).clone()
    res = {}
    for mode in ("exact", "two_stage"):
        eng.encode_mode = mode
        for _ in range(3): eng.encode_codes(frames)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out = eng.encode_codes(frames)
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / 10
        if mode == "two_stage":
            print("equal to exact:", bool(torch.equal(out, exact)), eng.encode_refine_stats(frames))
    print(f"frames {frames}: exact {res['exact']:.3f} ms, two-stage {res['two_stage']:.3f} ms")
    # the two stages by themselves (the list of the last run stays on the device: stage 2 alone repeats the same work)
    import ctypes as C
    b = eng._workspace(frames)
    gp, bias, tab, tau, _ = eng._two_stage
    d = L.EncodeDesc()
    d.n, d.h, d.w, d.levels, d.kc, d.segs = frames, eng.fh, eng.fw, eng.enc_levels, eng.kc, 1
    d.in_zx, d.in_delta = int(eng.shrink1.out_q[1]), float(eng.shrink1.out_q[0])
    codes = b["codes"]
    s1 = lambda: L.check(eng.lib.qv2x_codebook_encode_candidates_i8(C.byref(d), L.ptr(b["s1"]), L.ptr(gp), L.ptr(bias), L.ptr(tab), tau, L.ptr(codes),
                                                                    L.ptr(b["enc_list"]), L.ptr(b["enc_counters"]), L.current_stream()), "s1")
    s2 = lambda: L.check(eng.lib.qv2x_codebook_encode_listed_f32(C.byref(d), L.ptr(b["s1"]), eng.level_ptrs, L.ptr(b["enc_list"]), L.ptr(b["enc_counters"]),
                                                                 L.ptr(codes), L.current_stream()), "s2")
    s1(); torch.cuda.synchronize()
    print(f"stage 1 {bench.event_time_us(s1, 20):.1f} us, stage 2 {bench.event_time_us(s2, 20):.1f} us")

if not (len(sys.argv) > 2 and sys.argv[2] == "levels"):
    main()


def by_levels():
    """stage 1 alone with 1, 2, 3 levels (same operands: the kernel walks fewer tiles)"""
    import ctypes as C
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    state, eng, _, _ = bench.build_engine(min(32, os.cpu_count() or 8))
    _, full, _, _ = bench.frame_batch(1, 0, frames, torch.device("cuda"))
    eng(full)
    b = eng._workspace(frames)
    gp, bias, tab, tau, t = eng._two_stage
    for lv in (1, 2, 3):
        d = L.EncodeDesc()
        d.n, d.h, d.w, d.levels, d.kc, d.segs = frames, eng.fh, eng.fw, lv, eng.kc, 1
        d.in_zx, d.in_delta = int(eng.shrink1.out_q[1]), float(eng.shrink1.out_q[0])
        s1 = lambda: L.check(eng.lib.qv2x_codebook_encode_candidates_i8(C.byref(d), L.ptr(b["s1"]), L.ptr(gp), L.ptr(bias), L.ptr(tab), tau,
                                                                        L.ptr(b["codes"]), L.ptr(b["enc_list"]), L.ptr(b["enc_counters"]), L.current_stream()), "s1")
        print(f"levels {lv}: stage 1 {bench.event_time_us(s1, 20):.1f} us", flush=True)


if len(sys.argv) > 2 and sys.argv[2] == "levels":
    by_levels()


def empty_list():
    """stage 2 on an EMPTY list (counters zeroed): what the two launches cost by themselves"""
    import ctypes as C
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    state, eng, _, _ = bench.build_engine(min(32, os.cpu_count() or 8))
    _, full, _, _ = bench.frame_batch(1, 0, frames, torch.device("cuda"))
    eng(full); eng.encode_mode = "two_stage"; eng.encode_codes(frames); torch.cuda.synchronize()
    b = eng._workspace(frames)
    d = L.EncodeDesc()
    d.n, d.h, d.w, d.levels, d.kc, d.segs = frames, eng.fh, eng.fw, eng.enc_levels, eng.kc, 1
    d.in_zx, d.in_delta = int(eng.shrink1.out_q[1]), float(eng.shrink1.out_q[0])
    s2 = lambda: L.check(eng.lib.qv2x_codebook_encode_listed_f32(C.byref(d), L.ptr(b["s1"]), eng.level_ptrs, L.ptr(b["enc_list"]), L.ptr(b["enc_counters"]),
                                                                 L.ptr(b["codes"]), L.current_stream()), "s2")
    print(f"frames {frames}: stage 2 on the real list {bench.event_time_us(s2, 20):.1f} us", end="")
    b["enc_counters"].zero_(); torch.cuda.synchronize()
    print(f", on an EMPTY list {bench.event_time_us(s2, 20):.1f} us (inside a graph {bench.event_time_us(bench._graph_of(s2), 20):.1f})")


if len(sys.argv) > 2 and sys.argv[2] == "empty":
    empty_list()
