"""Pillar voxelizer contract: numpy generator (synth) and GPU kernel against the plain-Python statement in oracle/."""
import numpy as np
import pytest

from oracle.voxelize import voxelize as ref_voxelize
from quantv2x_amd import synth

RANGE, VS = synth.SHAPES["tiny"][0], synth.SHAPES["tiny"][1]


def _cloud(n, seed, sigma=6.0):
    g = np.random.default_rng(seed)
    pts = np.stack([g.normal(0, sigma, n), g.normal(0, sigma / 2, n), g.uniform(-3.5, 1.5, n), g.uniform(0, 1, n)], 1)
    return pts.astype(np.float32)     # includes out-of-range points on purpose


@pytest.mark.parametrize("n,max_points,max_voxels", [(500, 32, 2048), (5000, 32, 2048), (5000, 4, 100), (0, 32, 16)])
def test_numpy_voxelizer_follows_contract(n, max_points, max_voxels):
    pts = _cloud(max(n, 1), 1)[:n]
    f, c, m = synth.voxelize(pts, RANGE, VS, max_points, max_voxels) if n else ref_voxelize(pts, RANGE, VS, max_points, max_voxels)
    rf, rc, rm = ref_voxelize(pts, RANGE, VS, max_points, max_voxels)
    np.testing.assert_array_equal(f, rf); np.testing.assert_array_equal(c, rc); np.testing.assert_array_equal(m, rm)


@pytest.mark.gpu
@pytest.mark.parametrize("n,max_points,max_voxels", [(1, 32, 64), (700, 32, 2048), (6000, 32, 2048), (6000, 4, 100)])
def test_gpu_voxelizer_small(n, max_points, max_voxels):
    import torch
    from quantv2x_amd.voxelizer import GpuVoxelizer
    pts = _cloud(n, 2)
    rf, rc, rm = ref_voxelize(pts, RANGE, VS, max_points, max_voxels)
    f, c, m = GpuVoxelizer(RANGE, VS, max_points, max_voxels).one(torch.from_numpy(pts).cuda(), agent=3)
    np.testing.assert_array_equal(m.cpu().numpy(), rm)
    np.testing.assert_array_equal(c.cpu().numpy()[:, 1:], rc)
    assert (c.cpu().numpy()[:, 0] == 3).all()
    np.testing.assert_array_equal(f.cpu().numpy(), rf)


@pytest.mark.gpu
def test_gpu_voxelizer_empty_sweep():
    """a sweep with no point gives no pillar (and an agent without pillars is a valid model input)"""
    import torch
    from quantv2x_amd.voxelizer import GpuVoxelizer
    vox = GpuVoxelizer(RANGE, VS, 32, 64)
    f, c, m = vox.one(torch.empty((0, 4), dtype=torch.float32, device="cuda"), agent=0)
    assert f.shape == (0, 32, 4) and c.shape == (0, 4) and m.shape == (0,)
    both = vox([torch.empty((0, 4), dtype=torch.float32, device="cuda"), torch.from_numpy(_cloud(300, 4)).cuda()])
    assert (both["voxel_coords"][:, 0] == 1).all() and both["voxel_features"].shape[0] > 0


@pytest.mark.gpu
def test_gpu_voxelizer_full_size_feeds_the_model_inputs():
    """60k-point V2X-Real sweeps, two agents: identical to the numpy generator the rest of the suite uses."""
    import torch
    from quantv2x_amd.voxelizer import GpuVoxelizer
    rng, vs, max_vox, _ = synth.SHAPES["v2xreal"]
    sweeps = [synth.make_points(rng, 60000, 3000 + a, 35.0) for a in range(2)]
    got = GpuVoxelizer(rng, vs, 32, max_vox)([torch.from_numpy(s).cuda() for s in sweeps])
    want = synth.make_scene("v2xreal", n_agents=2, seed=3, n_points=60000)["inputs_m1"]
    for k in ("voxel_features", "voxel_coords", "voxel_num_points"):
        np.testing.assert_array_equal(got[k].cpu().numpy(), want[k], err_msg=k)
