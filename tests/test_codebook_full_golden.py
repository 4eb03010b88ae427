"""Index parity against the REFERENCE at scale: 35 200 rows (one V2X-Real agent-frame's worth) x 3 levels -- tests/golden/codebook_full.npz,
made by ``make_golden.py codebook_full`` from ``UMGMQuantizer.encode`` / ``_distance`` (codebook.py:106-131, 231-239, 330-337).

The rows are real encoder inputs (uint8 shrinker output of the integer path on the full-size synthetic frame): every fourth cell's row is
stored, the rest are those rows with their channels rotated by 64 / 128 / 192 -- exact on every host, unlike re-deriving a full-size PTQ
state (torch's BN folding and fp32 convolutions differ in the last bit between hosts).  CPU: the oracle's indices; ``-m gpu``: the exact
HIP kernel and the opt-in collapsed kernel on the same rows, the three-way table written to ``gpurun_out/`` (kept under ``profiles/``)."""
import copy
import json
import os

import numpy as np
import pytest

from _common import calibrated_plugin

TAU = 1e-4          # entries whose two best distances are closer than this are "fragile": fp32 summation order may pick either
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def golden_rows(g):
    codes_u8 = np.concatenate([np.roll(g["rows_u8_real"], 64 * j, axis=1) for j in range(4)])
    assert codes_u8.shape == (int(g["rows"]), 256) == (35200, 256)
    rows = (codes_u8.astype(np.float32) - np.float32(g["in_zp"])) * np.float32(g["in_delta"])
    return codes_u8, rows


def _state():
    from quantv2x_amd.ptq_state import export_ptq_state
    return export_ptq_state(calibrated_plugin())       # the codebook's parameters are the same at every grid size (seeded by name)


def _table(name, codes, g):
    mism = codes != g["codes"]
    solid = g["gaps"] > TAU
    return {"kernel": name, "rows": int(codes.shape[1]), "levels": int(codes.shape[0]),
            "mismatches": int(mism.sum()), "mismatches_with_gap_above_tau": int((mism & solid).sum()),
            "fragile_entries_gap_below_tau": int((~solid).sum()), "mismatches_among_fragile": int((mism & ~solid).sum()),
            "largest_gap_at_a_mismatch": float(g["gaps"][mism].max()) if mism.any() else 0.0, "tau": TAU}


def test_oracle_indices_vs_reference_at_scale(golden):
    from oracle.spec import Oracle
    g = golden["codebook_full"]
    _, rows = golden_rows(g)
    codes, gaps = Oracle(_state()).encode_rows(rows, want_gaps=True)
    t = _table("oracle (CPU restatement)", codes, g)
    print("codebook_full:", t)
    assert t["mismatches_with_gap_above_tau"] == 0, t                 # exact wherever the reference's own decision is not a coin toss
    assert t["fragile_entries_gap_below_tau"] >= 1                     # the rule is exercised: the sample HAS fragile entries
    assert t["mismatches"] <= t["fragile_entries_gap_below_tau"]
    agree = codes == g["codes"]
    # (a sanity check only: distances here reach 1e4-1e5, where one fp32 ulp is 1e-3 .. 8e-3 -- and still no index moves outside TAU)
    np.testing.assert_allclose(gaps[agree], g["gaps"][agree], rtol=0, atol=0.05)


@pytest.mark.gpu
def test_hip_exact_and_collapsed_indices_vs_reference_at_scale(golden):
    import torch
    from oracle.spec import Oracle
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    g = golden["codebook_full"]
    codes_u8, rows = golden_rows(g)
    # an engine of V2X-Real map size (100 x 352 = 35 200 cells) around the seeded weights; the encoder's input quantizer is the file's
    state = copy.copy(_state())
    state["meta/grid"] = np.array(synth.grid_size(*synth.SHAPES["v2xreal"][:2]), dtype=np.int64)
    state["shrinker_m1.layers.0.double_conv.1/a_delta"] = np.float32(g["in_delta"])
    state["shrinker_m1.layers.0.double_conv.1/a_zp"] = np.float32(g["in_zp"])
    eng = deploy(state=state)
    b = eng._workspace(1)
    assert (eng.fh, eng.fw) == (100, 352)
    b["s1"][0, 1:-1, 1:-1, :] = torch.from_numpy((codes_u8.astype(np.int16) - 128).astype(np.int8).reshape(100, 352, 256)).cuda()
    exact = eng.encode_codes(1).cpu().numpy().reshape(3, -1).copy()
    eng.encode_mode = "collapsed"
    collapsed = eng.encode_codes(1).cpu().numpy().reshape(3, -1).copy()
    torch.cuda.synchronize()
    orc_codes = Oracle(state).encode_rows(rows)
    table = [_table("oracle (CPU restatement)", orc_codes, g), _table("qv2x_codebook_encode_f32 (exact, default)", exact, g),
             _table("qv2x_codebook_encode_collapsed_f32 (opt-in)", collapsed, g)]
    table.append({"kernel": "exact HIP kernel vs oracle", "mismatches": int((exact != orc_codes).sum())})
    table.append({"kernel": "collapsed HIP kernel vs exact HIP kernel", "mismatches": int((collapsed != exact).sum()),
                  "gaps_at_those_entries": [float(x) for x in np.sort(g["gaps"][collapsed != exact])[:40]]})
    out = {"what": "codebook indices of 35 200 rows x 3 levels (real shrinker-output rows of a V2X-Real frame) against the reference's "
                   "UMGMQuantizer.encode on the same rows", "reference_thread_order_flips": int(g["codes_single_thread_differs"]), "table": table}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "codebook_index_parity_vs_reference.json"), "w") as f:
        json.dump(out, f, indent=1)
    assert np.array_equal(exact, orc_codes)                            # the kernel IS the restatement, bit for bit
    assert table[1]["mismatches_with_gap_above_tau"] == 0
    # the collapsed path is a different (float64-derived) algebra: where it leaves the reference, the reference's own top-2 gap must be
    # within the rounding noise of distances of 1e4-1e5 (one fp32 ulp there: up to 8e-3)
    assert table[2]["largest_gap_at_a_mismatch"] < 0.05, table[2]
