"""Per-kernel timings of the deployed path at full size (dev tool, runs on the GPU box).
    python tools/bench_kernels.py [conv|encode|all] [n_agents]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantv2x_amd import synth
from quantv2x_amd import lib as _L
if os.environ.get("QV2X_LIB_TAG"):           # a side library from tools/build_variant.py
    _L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "abl", f"libqv2x_{os.environ['QV2X_LIB_TAG']}.so")
from quantv2x_amd.engine import deploy

what = sys.argv[1] if len(sys.argv) > 1 else "all"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
chains = (sys.argv[3] != "nochain") if len(sys.argv) > 3 else True
eng = deploy(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "v2xreal_state.npz"))
import bench
dd = bench.frame_batch(1, 0, n, torch.device("cuda", 0))[1]          # n single-agent frames (a batch), as bench.py builds them
eng.use_chains = chains
eng(dd); torch.cuda.synchronize()


def timeit(fn, iters=30):
    """device time per call: `iters` launches captured into one HIP graph (no host launch overhead in the number)"""
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


if what in ("conv", "all"):
    tot_us, tot_ops = 0, 0
    for (kind, layer, x, h, w, out, c0, macs) in eng.conv_plan(n):
        if kind == "chain":
            us = timeit(lambda: eng._chain(layer, x, n, h, w, out))
            unit = "TOPS"
        elif kind == "conv":
            us = timeit(lambda: eng._conv(layer, x, n, h, w, out))
            unit = "TOPS"
        else:
            us = timeit(lambda: eng._deconv(layer, x, n, h, w, out, c0))
            unit = "TFLOPS"
        tot_us += us; tot_ops += 2 * macs if kind != "deconv" else 0
        print(f"{kind:6s} {layer.name:42s} h={h:4d} w={w:4d} macs={macs/1e9:7.3f}G  {us:8.1f} us  {2*macs/us/1e6:8.1f} {unit}")
    print(f"total {tot_us:.1f} us; conv {tot_ops/1e9:.1f} GOP")
if what in ("deconv", "conv", "all"):                                 # the three deblocks as the engine launches them: one batch
    items = [(layer, x, h, w, out, c0) for (kind, layer, x, h, w, out, c0, macs) in eng.conv_plan(n) if kind == "deconv"]
    flops = sum(2 * macs for (kind, *_r, macs) in eng.conv_plan(n) if kind == "deconv")
    us = timeit(lambda: eng._deconv_batch(items, n))
    print(f"deconv batch of {len(items)}: {us:.1f} us  ({flops/us/1e6:.1f} TFLOPS fp32 = {flops/us/1e6/157.3:.3f})")
if what in ("encode", "all"):
    us = timeit(lambda: eng.encode_codes(n), 10)
    print(f"encode {us:.1f} us  ({2*21.92*n/us*1e3:.1f} TFLOPS fp32)")
if what == "heads":                                                  # a11 on a fused map in memory (the general path's qv2x_heads_f32)
    hw = eng.fh * eng.fw
    rows = torch.randn((n, hw, 256), dtype=torch.float32, device="cuda")
    us = timeit(lambda: eng._run_heads(eng.heads, rows, n, hw))
    fl = 2.0 * n * hw * 256 * eng.heads.cout
    print(f"heads on {n} fused maps: {us:.1f} us  ({fl/us/1e6:.1f} TFLOPS on the {eng.heads.cout} real channels = {fl/us/1e6/157.3:.3f}; on the {eng.heads.cout_pad} padded ones {fl/us/1e6/157.3*eng.heads.cout_pad/eng.heads.cout:.3f})")
if what == "pfn":
    inp = dd["inputs_m1"]
    def run():
        eng.pillars_to_canvas(inp, n, resident=True); eng.clear_pillars(inp, n)
    run()
    print(f"pfn scatter + clear, {n} frames: {timeit(run):.1f} us")
    def run1():
        eng._workspace(n)["canvas_clean"] = True; eng.pillars_to_canvas(inp, n, resident=True)   # (dev: scatter only, onto a canvas it believes clean)
    try:
        run1(); print(f"pfn scatter alone: {timeit(run1):.1f} us")
    except Exception as e:
        print("scatter-alone timing skipped:", e)
    eng.clear_pillars(inp, n)
if what in ("rest", "all"):
    print("pfn", timeit(lambda: eng.pillars_to_canvas(dd["inputs_m1"], n)))
    hw = eng.fh * eng.fw
    codes = eng._workspace(n)["codes"]; pw = dd["pairwise_t_matrix"][0].contiguous()
    fused = torch.empty((1, hw, 256), dtype=torch.float32, device="cuda")
    from quantv2x_amd import lib as L
    print("fuse", timeit(lambda: eng.fuse(L.ptr(codes), hw, n * hw, None, pw, n, fused[0], 0)))
    print("heads", timeit(lambda: eng._run_heads(eng.heads, fused, 1, hw)))
    print("decode_rows", timeit(lambda: eng.decode_rows(codes, n * hw)))
    t0 = time.perf_counter(); 
    for _ in range(50): eng(dd)
    torch.cuda.synchronize(); print("eager frame ms", (time.perf_counter() - t0) / 50 * 1e3)
if what in ("rest", "all"):
    print("decode+single heads (one launch)", timeit(lambda: eng._decode_heads_single(codes, n)))
