"""The OPT-IN collapsed codebook encode against the exact one: indices may only differ where the exact path's two best distances are
within rounding error of each other (the exact kernel is the parity configuration; tests/test_hip_parity.py pins it)."""
import ctypes as C

import numpy as np
import pytest
import torch

from _common import calibrated_plugin, scene_np

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,n_agents,n_points", [("tiny", 2, 3000), ("small", 3, 12000), ("v2xreal", 1, 60000)])
def test_collapsed_indices_differ_only_at_ties(shape, n_agents, n_points):
    from oracle.spec import Oracle
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_plugin(shape, n_agents=n_agents, n_points=n_points))
    eng = deploy(state=state, device="cuda:0")
    sc = synth.scene_to_torch(scene_np(n_agents, shape, n_points=n_points), "cuda")
    exact = eng.encode_agents(sc["inputs_m1"], n_agents).clone()
    eng.encode_mode = "collapsed"
    coll = eng.encode_agents(sc["inputs_m1"], n_agents).clone()
    torch.cuda.synchronize()
    e, c = exact.cpu().numpy().reshape(eng.levels, -1), coll.cpu().numpy().reshape(eng.levels, -1)
    b = eng._workspace(n_agents)
    s1 = (b["s1"][:, 1:-1, 1:-1, :].to(torch.int16) + 128).cpu().numpy().astype(np.float32)
    q = eng.shrink1.out_q
    want, gaps = Oracle(state).encode_rows(((s1 - np.float32(q[1])) * np.float32(q[0])).reshape(-1, 256), want_gaps=True)
    assert np.array_equal(e, want)
    mm = e != c
    assert mm.any(axis=0).mean() < 1e-3
    seen = np.zeros(mm.shape[1], bool)
    for l in range(eng.levels):                                      # the first differing level of a cell is the rounding event
        first = mm[l] & ~seen
        seen |= mm[l]
        if first.any():
            assert gaps[l][first].max() < 0.25 * np.percentile(gaps[l], 1), (l, gaps[l][first].max(), np.percentile(gaps[l], 1))


def test_collapsed_mode_runs_the_whole_frame_and_can_be_switched_back():
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_plugin("tiny", n_agents=2))
    eng = deploy(state=state, device="cuda:0")
    sc = synth.scene_to_torch(scene_np(2), "cuda")
    a = eng(sc)["preds_tensor"].clone()
    eng.encode_mode = "collapsed"
    b = eng(sc)["preds_tensor"].clone()
    eng.encode_mode = "exact"
    c = eng(sc)["preds_tensor"].clone()
    assert torch.equal(a, c) and torch.isfinite(b).all() and (a != b).float().mean() < 0.05      # a flipped index moves a few cells' predictions
    eng.encode_mode = "nonsense"
    with pytest.raises(ValueError):
        eng(sc)


def test_argument_checks():
    from quantv2x_amd import lib as L
    lib = L.load()
    d = L.EncodeDesc()
    d.n, d.h, d.w, d.levels, d.kc, d.in_zx, d.in_delta = 1, 16, 32, 4, 128, 0, 0.1
    p = C.c_void_p(256)
    assert lib.qv2x_codebook_encode_collapsed_f32(C.byref(d), p, p, p, p, p, None) == -1 and b"levels" in lib.qv2x_last_error()
    d.levels, d.kc = 3, 48
    assert lib.qv2x_codebook_encode_collapsed_f32(C.byref(d), p, p, p, p, p, None) == -1
    d.kc = 128
    assert lib.qv2x_codebook_encode_collapsed_f32(C.byref(d), p, p, p, None, p, None) == -1 and b"tables" in lib.qv2x_last_error()
    assert lib.qv2x_codebook_encode_collapsed_f32(C.byref(d), C.c_void_p(8), p, p, p, p, None) == -2
