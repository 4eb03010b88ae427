"""The two-stage exact encode's THEORY on the CPU (quantv2x_amd/encode_two_stage.py): stage 1 -- the collapsed scores in exact integer
arithmetic, emulated in numpy -- accepts a cell only when its gap at every level exceeds a bound that provably covers the fp32 chain's
rounding; every accepted cell must then carry the oracle's indices (the reference-order fp32 chain, oracle/qv2x_oracle.c) at every level.
Checked on the reference's own 35 200 golden rows (tests/golden/codebook_full.npz: against the oracle AND against the indices
``UMGMQuantizer.encode`` itself produced, codebook.py:106-131, 231-239, 330-337), on a codebook with DUPLICATED and NEARLY duplicated
codewords (exact ties -- first index wins in the reference, codebook.py:110 ``argmin`` -- and gaps spread around the bound), and on other
dictionary sizes.  The HIP kernels are held to the same statements in tests/test_hip_encode_two_stage.py."""
import copy

import numpy as np
import pytest

from test_codebook_full_golden import _state, golden_rows


def _levels_kc(state):
    levels = sum(1 for l in range(4) if f"codebook/{l}/codebook" in state)
    return levels, int(state["codebook/0/codebook"].shape[0])


DUPLICATES = {5: 3, 70: 9, 40: 2, 41: 1, 90: 0, 100: 50, 101: 60, 110: 30, 111: 31, 120: 64}


def adversarial_state(state):
    """Duplicated codewords (exact ties in every kernel: c_5 = c_3, c_70 = c_9, ... at every level) and near-duplicates at distances that put
    the two best scores 1e-4 .. 1 apart (below, around and above the bound)."""
    st = copy.copy(state)
    rng = np.random.default_rng(5)
    for l in range(3):
        cb = np.array(st[f"codebook/{l}/codebook"], dtype=np.float32, copy=True)
        for dup, src in DUPLICATES.items():
            cb[dup] = cb[src]
        for k, src, eps in ((20, 11, 1e-5), (21, 12, 1e-3), (22, 13, 1e-2), (23, 14, 3e-2), (24, 15, 0.1), (25, 16, 0.3)):
            v = rng.standard_normal(256).astype(np.float32)
            cb[k] = cb[src] + np.float32(eps) * v / np.float32(np.linalg.norm(v))
        st[f"codebook/{l}/codebook"] = cb
    return st


def check_theory(state, codes_u8, in_delta, in_zp, ref_codes=None, name=""):
    from oracle.spec import Oracle
    from quantv2x_amd.encode_two_stage import candidate_emulate, candidate_tables
    levels, kc = _levels_kc(state)
    tabs = candidate_tables(state, levels, float(in_delta), int(in_zp))
    cand, flagged, flags, gaps = candidate_emulate(codes_u8, tabs, float(in_delta), int(in_zp))
    rows = (codes_u8.astype(np.float32) - np.float32(in_zp)) * np.float32(in_delta)
    want, ogaps = Oracle(state).encode_rows(rows, want_gaps=True)
    ok = ~flagged
    wrong = (cand != want).any(0)
    print(f"two-stage theory [{name}]: {len(ok)} rows, flagged {flagged.mean():.4f} (per level {flags.mean(1).round(4).tolist()}), "
          f"candidate != oracle on {int(wrong.sum())} rows, of them accepted {int((wrong & ok).sum())}; "
          f"tau at |x| = 134: {[float(t[0] + 134 * t[1] + 134 * 134 * t[2]) * tabs['h'] for t in tabs['tau']]}")
    assert not (wrong & ok).any(), "an ACCEPTED cell differs from the fp32 chain: the bound does not hold"
    # accepted cells have a strict minimum in the fp32 chain too: no exact tie among them
    assert (ogaps[:, ok] > 0).all()
    if ref_codes is not None:
        assert not ((cand != ref_codes).any(0) & ok).any(), "an accepted cell differs from the reference's own index"
    return flagged, cand, want, ogaps


def test_golden_rows_accepted_cells_equal_the_oracle_and_the_reference(golden):
    g = golden["codebook_full"]
    codes_u8, _ = golden_rows(g)
    flagged, cand, want, _ = check_theory(_state(), codes_u8, g["in_delta"], g["in_zp"], ref_codes=g["codes"], name="reference golden rows")
    assert 0.01 < flagged.mean() < 0.35                                  # the bound is neither vacuous nor useless
    assert ((cand != want).any(0)).sum() >= 1                           # the sample has cells where real arithmetic and the fp32 chain disagree


def test_duplicated_and_nearly_duplicated_codewords(golden):
    g = golden["codebook_full"]
    codes_u8, _ = golden_rows(g)
    codes_u8 = codes_u8[::4]
    st = adversarial_state(_state())
    flagged, cand, want, ogaps = check_theory(st, codes_u8, g["in_delta"], g["in_zp"], name="duplicated / nearly duplicated codewords")
    # exact ties exist, the fp32 chain resolves them to the FIRST index, and every one of them is sent to stage 2
    ties = (ogaps == 0).any(0)
    assert ties.sum() > 100 and flagged[ties].all()
    assert np.isin(want, list(DUPLICATES)).sum() == 0 and np.isin(want, list(DUPLICATES.values())).sum() > 100
    # gaps on both sides of the bound occur: some near-duplicates are accepted, some flagged
    near = np.isin(want, [11, 12, 13, 14, 15, 16, 20, 21, 22, 23, 24, 25]).any(0)
    assert (flagged & near).sum() > 10 and (~flagged & near).sum() > 10


@pytest.mark.parametrize("kc", [64, 32, 96])
def test_other_dictionary_sizes(golden, kc):
    g = golden["codebook_full"]
    codes_u8, _ = golden_rows(g)
    st = copy.copy(_state())
    for l in range(3):
        st[f"codebook/{l}/codebook"] = np.ascontiguousarray(st[f"codebook/{l}/codebook"][:kc])
    check_theory(st, codes_u8[::8], g["in_delta"], g["in_zp"], name=f"dict {kc}")


def test_refusals_and_the_grid():
    from quantv2x_amd.encode_two_stage import G_MAX_INT, candidate_tables
    st = _state()
    tabs = candidate_tables(st, 3, 0.387, 0)
    gi = tabs["g_int"]
    assert np.abs(gi).max() == G_MAX_INT and tabs["gpack"].shape == (3, 4, 3, 8, 64, 16)
    # the three limbs put back together are the grid values, in the fragment order the kernel reads
    p = tabs["gpack"].astype(np.int64)
    lane, byte = 37, 9                                                  # half 1, score 5 of its tile
    for lv, t, s in ((0, 0, 0), (2, 3, 7), (1, 2, 4)):
        v = p[lv, t, 0, s, lane, byte] + 256 * p[lv, t, 1, s, lane, byte] + 65536 * p[lv, t, 2, s, lane, byte]
        assert v == gi[lv * 128 + 32 * t + (lane & 31), 32 * s + 16 * (lane >> 5) + byte]
    big = copy.copy(st)
    big["codebook/0/codebook"] = np.concatenate([st["codebook/0/codebook"], st["codebook/0/codebook"]])
    with pytest.raises(ValueError):
        candidate_tables(big, 3, 0.387, 0)
