"""The two-stage EXACT encode on the GPU (csrc/codebook_encode_cand.hip + the LIST form of codebook_encode_wave_kernel; theory and operands:
quantv2x_amd/encode_two_stage.py, pinned on the CPU by tests/test_encode_two_stage_cpu.py).  Gates:

  * ``encode_mode = "two_stage"`` gives the indices of the reference-order kernel AND of the oracle, bit for bit -- on the reference's own
    35 200 golden rows (also against ``UMGMQuantizer.encode``'s indices themselves, codebook.py:106-131, 231-239, 330-337), on a codebook
    with duplicated / nearly duplicated codewords (exact ties: the first index, as ``argmin`` at codebook.py:110), on ragged launches,
    other dictionary sizes, several frames, and replayed from a HIP graph on fresh input;
  * stage 1 alone equals its numpy emulation bit for bit: the candidate indices of EVERY cell and the set of listed cells (so the CPU
    proof obligations are statements about what runs here);
  * the refine fraction is the CPU's."""
import copy

import numpy as np
import pytest
import torch

from test_codebook_full_golden import _state, golden_rows
from test_encode_two_stage_cpu import adversarial_state

pytestmark = pytest.mark.gpu


def v2xreal_engine(state, g):
    """an engine of V2X-Real map size (100 x 352 = 35 200 cells) around the seeded weights; the encoder's input quantizer is the file's"""
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    st = copy.copy(state)
    st["meta/grid"] = np.array(synth.grid_size(*synth.SHAPES["v2xreal"][:2]), dtype=np.int64)
    st["shrinker_m1.layers.0.double_conv.1/a_delta"] = np.float32(g["in_delta"])
    st["shrinker_m1.layers.0.double_conv.1/a_zp"] = np.float32(g["in_zp"])
    eng = deploy(state=st)
    eng._workspace(1)
    assert (eng.fh, eng.fw) == (100, 352)
    return st, eng


def put_rows(eng, n, codes_u8):
    b = eng._workspace(n)
    b["s1"][:, 1:-1, 1:-1, :] = torch.from_numpy((codes_u8.astype(np.int16) - 128).astype(np.int8).reshape(n, eng.fh, eng.fw, 256)).cuda()
    return b


def run_modes(eng, n):
    out = {}
    for mode in ("exact", "two_stage"):
        eng.encode_mode = mode
        out[mode] = eng.encode_codes(n).cpu().numpy().reshape(eng.levels, -1).copy()
    torch.cuda.synchronize()
    return out


def stage1_alone(eng, n):
    """qv2x_codebook_encode_candidates_i8 by itself: (codes [levels, M], sorted list of listed cells, counters)"""
    import ctypes as C
    from quantv2x_amd import lib as L
    b = eng._workspace(n)
    eng.encode_mode = "two_stage"
    eng.encode_codes(n)                                                  # builds the operands and the buffers
    gp, bias, tab, tau, _ = eng._two_stage
    d = L.EncodeDesc()
    d.n, d.h, d.w, d.levels, d.kc, d.segs = n, eng.fh, eng.fw, eng.enc_levels, eng.kc, 1
    d.in_zx, d.in_delta = int(eng.shrink1.out_q[1]), float(eng.shrink1.out_q[0])
    codes = torch.zeros((eng.levels, n * eng.fh * eng.fw), dtype=torch.uint8, device="cuda")
    L.check(eng.lib.qv2x_codebook_encode_candidates_i8(C.byref(d), L.ptr(b["s1"]), L.ptr(gp), L.ptr(bias), L.ptr(tab), tau, L.ptr(codes),
                                                       L.ptr(b["enc_list"]), L.ptr(b["enc_counters"]), L.current_stream()), "candidates")
    torch.cuda.synchronize()
    cnt = b["enc_counters"].cpu().numpy().astype(np.int64)
    m_cells = n * eng.fh * eng.fw
    lists = [b["enc_list"][c * m_cells:c * m_cells + cnt[1 + c]].cpu().numpy().astype(np.int64) for c in range(3)]     # by first undecided level
    assert cnt[0] == sum(len(v) for v in lists)
    return codes.cpu().numpy(), np.sort(np.concatenate(lists)), cnt, lists


@pytest.mark.parametrize("adversarial", [False, True])
def test_golden_rows_two_stage_equals_exact_oracle_and_reference(golden, adversarial):
    from oracle.spec import Oracle
    from quantv2x_amd.encode_two_stage import candidate_emulate
    g = golden["codebook_full"]
    codes_u8, rows = golden_rows(g)
    base = adversarial_state(_state()) if adversarial else _state()
    st, eng = v2xreal_engine(base, g)
    put_rows(eng, 1, codes_u8)
    got = run_modes(eng, 1)
    want = Oracle(st).encode_rows(rows)
    assert np.array_equal(got["exact"], want)
    mism = int((got["two_stage"] != want).sum())
    assert mism == 0, f"two-stage encode: {mism} indices differ from the oracle"
    if not adversarial:
        assert np.array_equal(got["two_stage"], g["codes"]) or int((got["two_stage"] != g["codes"]).sum()) == int((want != g["codes"]).sum())
    # stage 1 alone = the CPU emulation, bit for bit: every candidate index, and exactly the same cells listed
    cand, listed, cnt, lists = stage1_alone(eng, 1)
    t = eng._two_stage[4]
    e_cand, e_flag, e_flags, _ = candidate_emulate(codes_u8, t, float(g["in_delta"]), int(g["in_zp"]))
    assert np.array_equal(listed, np.flatnonzero(e_flag)), (len(listed), int(e_flag.sum()))
    ok = ~e_flag
    assert np.array_equal(cand[:, ok], e_cand[:, ok])
    # (a listed cell's later levels follow a candidate the kernel and the emulation both computed: equal as well)
    assert np.array_equal(cand, e_cand)
    first = [e_flags[l] & ~e_flags[:l].any(0) for l in range(3)]
    assert cnt[0] == e_flag.sum() and list(cnt[1:4]) == [int(f.sum()) for f in first]
    for c in range(3):                                               # list c = exactly the cells first undecided at level c
        assert np.array_equal(np.sort(lists[c]), np.flatnonzero(first[c])), c
    stats = eng.encode_refine_stats(1)
    print("two-stage encode on the golden rows:", stats, "adversarial" if adversarial else "")
    assert 0.01 < stats["refined_fraction"] < 0.5
    if adversarial:
        ties = np.isin(want, [3, 9, 2, 1, 0]).any(0)
        assert ties.sum() > 100


@pytest.mark.parametrize("kc", [64, 32, 96])
def test_other_dictionary_sizes(golden, kc):
    from oracle.spec import Oracle
    g = golden["codebook_full"]
    codes_u8, rows = golden_rows(g)
    base = copy.copy(_state())
    for l in range(3):
        base[f"codebook/{l}/codebook"] = np.ascontiguousarray(base[f"codebook/{l}/codebook"][:kc])
    st, eng = v2xreal_engine(base, g)
    put_rows(eng, 1, codes_u8)
    got = run_modes(eng, 1)
    want = Oracle(st).encode_rows(rows)
    assert np.array_equal(got["exact"], want) and np.array_equal(got["two_stage"], want)


@pytest.mark.parametrize("n_agents", [1, 2, 3])
def test_tiny_frames_ragged_launches(n_agents):
    """whole frames at tiny shape (512 cells per agent: less than one wave's 128 cells x 4 per workgroup in the last block, one to three
    frames) through the model call: the two-stage engine's output dict equals the exact engine's bit for bit, indices equal the oracle's"""
    from _common import calibrated_plugin, scene_np
    from oracle.spec import Oracle
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_plugin("tiny", n_agents=2))
    eng = deploy(state=state)
    sc = scene_np(n_agents, seed=13)
    dd = synth.scene_to_torch(sc, "cuda")
    eng.encode_mode = "exact"
    te = {}
    want = {k: v.clone() for k, v in eng(dd, te).items()}
    exact_codes = te["codes"].clone()
    eng.encode_mode = "two_stage"
    tt = {}
    got = eng(dd, tt)
    torch.cuda.synchronize()
    assert torch.equal(tt["codes"], exact_codes)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    ot = {}
    Oracle(state).forward(sc, ot)
    assert np.array_equal(tt["codes"].cpu().numpy().reshape(ot["codes"].shape), ot["codes"])


def test_graph_replay_on_fresh_rows(golden):
    """both launches (and the counter reset) inside ONE HIP graph: replayed on rows written after the capture, the indices follow the rows"""
    from oracle.spec import Oracle
    g = golden["codebook_full"]
    codes_u8, rows = golden_rows(g)
    st, eng = v2xreal_engine(_state(), g)
    eng.encode_mode = "two_stage"
    b = put_rows(eng, 1, codes_u8)
    eng.encode_codes(1)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        eng.encode_codes(1)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            out = eng.encode_codes(1)
    torch.cuda.synchronize()
    orc = Oracle(st)
    for shift in (0, 17, 101):
        fresh = np.roll(codes_u8, shift, axis=1)
        put_rows(eng, 1, fresh)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        want = orc.encode_rows((fresh.astype(np.float32) - np.float32(g["in_zp"])) * np.float32(g["in_delta"]))
        assert np.array_equal(out.cpu().numpy().reshape(3, -1), want), shift
        assert eng.encode_refine_stats(1)["refined"] > 0


def test_refusals():
    import ctypes as C
    from quantv2x_amd import lib as L
    lib = L.load()
    d = L.EncodeDesc()
    d.n, d.h, d.w, d.levels, d.kc, d.segs = 1, 4, 4, 3, 128, 1
    assert lib.qv2x_codebook_encode_candidates_i8(C.byref(d), None, None, None, None, None, None, None, None, None) != 0
    assert lib.qv2x_codebook_encode_listed_f32(C.byref(d), None, None, None, None, None, None) != 0
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    tau = (C.c_float * 9)()
    d.kc = 256
    assert lib.qv2x_codebook_encode_candidates_i8(C.byref(d), L.ptr(buf), L.ptr(buf), L.ptr(buf), L.ptr(buf), tau, L.ptr(buf), L.ptr(buf), L.ptr(buf), None) != 0
    d.kc, d.segs = 128, 2
    assert lib.qv2x_codebook_encode_candidates_i8(C.byref(d), L.ptr(buf), L.ptr(buf), L.ptr(buf), L.ptr(buf), tau, L.ptr(buf), L.ptr(buf), L.ptr(buf), None) != 0


@pytest.mark.parametrize("counts", [(31, 0, 0), (1000, 12000, 19730), (1024 * 32 - 40, 0, 33), (20000, 20000, 9152), (300, 49000, 137),
                                    (0, 0, 1536 * 32), (0, 0, 1536 * 32 + 1), (40, 511 * 32, 1024 * 32 - 7), (70400, 0, 0), (5, 7, 70388)])
def test_stage2_alone_on_constructed_lists(golden, counts):
    """qv2x_codebook_encode_listed_f32 by itself on lists of chosen lengths: the device-side split (whole rounds of persistent waves, every second
    round backwards; the remainder as workgroups; more than `list_tail_max` remainder tiles = one more round of waves) at its boundaries --
    1 tile; 2 x 1024 tiles - 1 (+ a second list); a remainder of exactly 512 and of 513 tiles; every cell of two frames listed -- with partial
    tiles in every list.  The listed cells' codes from their list's level on are overwritten with garbage first: stage 2 must restore the
    every-cell kernel's indices, using the STORED codes below that level."""
    import ctypes as C
    from quantv2x_amd import lib as L
    g = golden["codebook_full"]
    codes_u8, _ = golden_rows(g)
    st, eng = v2xreal_engine(_state(), g)
    n = 2
    rng = np.random.default_rng(5)
    rows = np.concatenate([codes_u8, np.roll(codes_u8, 7, axis=0)[:, rng.permutation(256)]])
    b = put_rows(eng, n, rows)
    eng.encode_mode = "exact"
    ref = eng.encode_codes(n).clone()                                   # [levels, M]
    torch.cuda.synchronize()
    m_cells = n * eng.fh * eng.fw
    assert sum(counts) <= m_cells
    eng.encode_mode = "two_stage"
    eng.encode_codes(n)                                                 # (allocates the list buffers)
    cells = torch.from_numpy(rng.permutation(m_cells)[:sum(counts)].astype(np.int32)).cuda()
    work = ref.clone().reshape(eng.levels, m_cells)
    lst, cnt, at = b["enc_list"], b["enc_counters"], 0
    lst.fill_(-1)
    for c, k in enumerate(counts):
        mine = cells[at:at + k]
        lst[c * m_cells:c * m_cells + k] = mine
        for l in range(c, eng.levels):
            work[l, mine.long()] = 255 - l
        at += k
    cnt.copy_(torch.tensor([sum(counts), *counts], dtype=torch.int32))
    d = L.EncodeDesc()
    d.n, d.h, d.w, d.levels, d.kc, d.segs = n, eng.fh, eng.fw, eng.enc_levels, eng.kc, 1
    d.in_zx, d.in_delta = int(eng.shrink1.out_q[1]), float(eng.shrink1.out_q[0])
    L.check(eng.lib.qv2x_codebook_encode_listed_f32(C.byref(d), L.ptr(b["s1"]), eng.level_ptrs, L.ptr(lst), L.ptr(cnt), L.ptr(work), L.current_stream()), "listed")
    torch.cuda.synchronize()
    bad = (work != ref.reshape(eng.levels, m_cells)).sum().item()
    assert bad == 0, f"{bad} indices differ after stage 2 on lists of {counts} cells"
