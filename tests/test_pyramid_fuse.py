"""weighted_fuse of the HEAL Pyramid fusion (first piece): numpy oracle and torch mirror against vectors from
the reference's own function (tests/golden/pyramid_fuse.npz), and the HIP kernel against the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import geometry

CASES = [(tag, name, n) for tag in ("l0", "l1") for name in ("near", "far") for n in (1, 2, 3)]


def _affine(g, name):
    pw = g["pairwise"] if name == "near" else g["pairwise_far"]
    return geometry.normalize_pairwise_tfm(pw, 12.8, 25.6, 1), pw


@pytest.mark.parametrize("tag,name,n", CASES)
def test_oracle_and_mirror_match_the_reference(golden, tag, name, n):
    from quantv2x_amd.plugin.models.fuse_modules.pyramid_fuse import weighted_fuse
    g = golden["pyramid_fuse"]
    aff, _ = _affine(g, name)
    x, score, want = g[f"{tag}/x"][:n], g[f"{tag}/score"][:n], g[f"{tag}/{name}_n{n}"][0]
    got = geometry.weighted_fuse(x.transpose(0, 2, 3, 1), score.transpose(0, 2, 3, 1), aff[0], n)
    np.testing.assert_allclose(got.transpose(2, 0, 1), want, rtol=1e-5, atol=2e-6)
    with torch.no_grad():
        mir = weighted_fuse(torch.from_numpy(x), torch.from_numpy(score.copy()), torch.tensor([n]), torch.from_numpy(aff), False)
    np.testing.assert_allclose(mir[0].numpy(), want, rtol=1e-6, atol=1e-6)
    if name == "far" and n == 3:                                      # the out-of-view agent contributes nothing anywhere
        np.testing.assert_allclose(want, g[f"{tag}/far_n2"][0], rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,name,n", CASES)
def test_hip_kernel_matches_the_oracle(golden, tag, name, n):
    from quantv2x_amd import lib as L
    lib = L.load()
    g = golden["pyramid_fuse"]
    aff, pw = _affine(g, name)
    x = np.ascontiguousarray(g[f"{tag}/x"][:n].transpose(0, 2, 3, 1))
    score = np.ascontiguousarray(g[f"{tag}/score"][:n].transpose(0, 2, 3, 1))
    _, h, w, c = x.shape
    want = geometry.weighted_fuse(x, score, aff[0], n)
    d = L.FuseDesc()
    d.agents, d.h, d.w, d.levels, d.kc, d.max_cav, d.ego = n, h, w, 1, 1, 5, 0
    d.h_metres, d.w_metres, d.discrete_ratio = 12.8, 25.6, 1.0
    xt, st = torch.from_numpy(x).cuda(), torch.from_numpy(score).cuda()
    pwt = torch.from_numpy(np.ascontiguousarray(pw[0])).cuda()
    out = torch.full((h * w, c), 7.0, dtype=torch.float32, device="cuda")
    L.check(lib.qv2x_pyramid_weighted_fuse_f32(C.byref(d), c, L.ptr(xt), L.ptr(st), L.ptr(pwt), L.ptr(out), L.current_stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy().reshape(h, w, c), want, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(out.cpu().numpy().reshape(h, w, c).transpose(2, 0, 1), g[f"{tag}/{name}_n{n}"][0], rtol=2e-5, atol=2e-5)
    d.agents = 9
    assert lib.qv2x_pyramid_weighted_fuse_f32(C.byref(d), c, L.ptr(xt), L.ptr(st), L.ptr(pwt), L.ptr(out), None) == -1
