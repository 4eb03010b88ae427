"""Index parity against the REFERENCE at scale: the 35 200 rows of one V2X-Real agent-frame (tests/golden/codebook_full.npz, made by
``make_golden.py codebook_full`` from ``UMGMQuantizer.encode`` / ``_distance``, codebook.py:106-131, 231-239, 330-337).

The input rows are not stored: they are the dequantized uint8 shrinker output of the integer path, regenerated here from seeds
(mirror -> W8A8 min-max state -> CPU oracle); the golden file holds their checksum, so a drift anywhere upstream fails loudly
instead of comparing codes of different rows.  CPU: the oracle's indices; ``-m gpu``: the exact HIP kernel and the opt-in collapsed
kernel, with the three-way mismatch table written to ``gpurun_out/`` (copied to ``profiles/``)."""
import json
import os

import numpy as np
import pytest

from _common import calibrated_plugin, scene_np

TAU = 1e-4          # rows whose two best distances are closer than this are "fragile": fp32 summation order may pick either
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_cache = {}


def _weight_checksums(code):
    c = code.reshape(-1).astype(np.int64)
    return np.array([c.sum(), (c * (1 + np.arange(c.size) % 251)).sum()], dtype=np.int64)


def fullsize_rows():
    """(state, scene, oracle taps) of the V2X-Real frame the golden file was made on"""
    if not _cache:
        from oracle.spec import Oracle
        from quantv2x_amd.ptq_state import export_ptq_state
        state = export_ptq_state(calibrated_plugin("v2xreal", n_agents=1, n_points=60000))
        sc = scene_np(1, "v2xreal", n_points=60000)
        taps = {}
        orc = Oracle(state)
        orc.forward(sc, taps)
        _cache.update(state=state, sc=sc, taps=taps, orc=orc)
    return _cache


def _table(name, codes, g):
    mism = codes != g["codes"]
    solid = g["gaps"] > TAU
    return {"kernel": name, "rows": int(codes.shape[1]), "levels": int(codes.shape[0]),
            "mismatches": int(mism.sum()), "mismatches_with_gap_above_tau": int((mism & solid).sum()),
            "fragile_entries_gap_below_tau": int((~solid).sum()), "mismatches_among_fragile": int((mism & ~solid).sum()),
            "largest_gap_at_a_mismatch": float(g["gaps"][mism].max()) if mism.any() else 0.0, "tau": TAU}


def test_oracle_indices_vs_reference_at_scale(golden):
    g = golden["codebook_full"]
    c = fullsize_rows()
    shr = c["taps"]["shrinker_m1.layers.0.double_conv.1"]
    np.testing.assert_array_equal(_weight_checksums(shr), g["in_checksum"], err_msg="the regenerated input rows are not the golden file's")
    dq, zq = c["taps"]["shrinker_q"]
    assert np.float32(dq) == g["in_delta"] and np.float32(zq) == g["in_zp"]
    rows = ((shr.astype(np.float32) - np.float32(zq)) * np.float32(dq)).reshape(-1, 256)
    assert rows.shape[0] == int(g["rows"]) == 35200
    codes, gaps = c["orc"].encode_rows(rows, want_gaps=True)
    t = _table("oracle (CPU restatement)", codes, g)
    assert t["mismatches_with_gap_above_tau"] == 0, t                 # exact wherever the reference's own decision is not a coin toss
    assert t["fragile_entries_gap_below_tau"] >= 1                     # the rule is exercised: the sample HAS fragile rows
    assert t["mismatches"] <= t["fragile_entries_gap_below_tau"]
    agree = codes == g["codes"]
    # (a sanity check only: distances here reach 1e4-1e5, where one fp32 ulp is 1e-3 .. 8e-3 -- and still no index moves outside TAU)
    np.testing.assert_allclose(gaps[agree], g["gaps"][agree], rtol=0, atol=0.05)
    print("codebook_full:", t)
    np.testing.assert_array_equal(c["taps"]["codes"].reshape(3, -1), codes)   # the forward pass's own indices are these


@pytest.mark.gpu
def test_hip_exact_and_collapsed_indices_vs_reference_at_scale(golden):
    import torch
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    g = golden["codebook_full"]
    c = fullsize_rows()
    eng = deploy(state=c["state"])
    dd = synth.scene_to_torch(c["sc"], "cuda")
    taps = {}
    eng(dd, taps)
    torch.cuda.synchronize()
    shr = (taps["shrinker_m1.layers.0.double_conv.1"][:, 1:-1, 1:-1].to(torch.int16) + 128).to(torch.uint8).cpu().numpy()
    np.testing.assert_array_equal(_weight_checksums(shr), g["in_checksum"])
    exact = taps["codes"].cpu().numpy().reshape(3, -1).copy()
    eng.encode_mode = "collapsed"
    collapsed = eng.encode_agents(dd["inputs_m1"], 1).cpu().numpy().reshape(3, -1).copy()
    torch.cuda.synchronize()
    orc_codes = c["taps"]["codes"].reshape(3, -1)
    rows = [_table("oracle (CPU restatement)", orc_codes, g), _table("qv2x_codebook_encode_f32 (exact, default)", exact, g),
            _table("qv2x_codebook_encode_collapsed_f32 (opt-in)", collapsed, g)]
    rows.append({"kernel": "exact HIP kernel vs oracle", "mismatches": int((exact != orc_codes).sum())})
    rows.append({"kernel": "collapsed HIP kernel vs exact HIP kernel", "mismatches": int((collapsed != exact).sum()),
                 "of_which_first_level_differs": int((collapsed[0] != exact[0]).sum())})
    out = {"what": "codebook indices of one V2X-Real agent-frame (35 200 rows x 3 levels) against the reference's UMGMQuantizer.encode on the same rows",
           "reference_thread_order_flips": int(g["codes_single_thread_differs"]), "table": rows}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "codebook_index_parity_vs_reference.json"), "w") as f:
        json.dump(out, f, indent=1)
    assert np.array_equal(exact, orc_codes)                            # the kernel IS the restatement, bit for bit
    assert rows[1]["mismatches_with_gap_above_tau"] == 0
    # the collapsed path: a different (float64-derived) algebra -- it may differ from the reference, but only at rows the reference itself
    # decides by rounding noise
    assert rows[2]["mismatches_with_gap_above_tau"] == 0, rows[2]
