"""The un-quantized model on the HIP path (``quantv2x_amd/engine_fp32.py``, ``csrc/fp32_path.hip``) against its CPU oracle
(``oracle/spec_fp32.py``): every fp32 activation of a1-a4 BIT-identical (one fma chain per output in a fixed order on both sides),
every codebook index identical, the fused map and the (un-quantized) predictions within fp32 tolerance."""
import numpy as np
import pytest
import torch

from _common import FUSE_TOL, build_plugin, scene_np

pytestmark = pytest.mark.gpu
torch.set_num_threads(8)


def _interior(t):
    return t[:, 1:-1, 1:-1, :].cpu().numpy()


def _compare(orc, eng, sc):
    from quantv2x_amd import synth
    ot, gt = {}, {}
    want = orc.forward(sc, ot)
    got = eng(synth.scene_to_torch(sc, "cuda"), gt)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(_interior(gt["canvas"]), ot["canvas"], err_msg="canvas")
    n = 0
    for name, arr in ot.items():
        if name.startswith("backbone_m1.blocks") or name.startswith("shrinker_m1"):
            np.testing.assert_array_equal(_interior(gt[name]) if name in gt else _interior(eng._workspace(arr.shape[0])["s1"]), arr, err_msg=name)
            n += 1
    assert n >= 20
    np.testing.assert_array_equal(_interior(gt["cat"]), ot["cat"], err_msg="cat")
    np.testing.assert_array_equal(gt["codes"].cpu().numpy().reshape(ot["codes"].shape), ot["codes"])
    h, w = ot["fused"].shape[1:3]
    np.testing.assert_allclose(gt["fused"].cpu().numpy().reshape(-1, h, w, 256), ot["fused"], **FUSE_TOL)
    for k in want:
        np.testing.assert_allclose(got[k].cpu().numpy(), want[k], rtol=2e-4, atol=2e-4, err_msg=k)
    return got


@pytest.mark.parametrize("n_agents", [1, 3])
def test_tiny_fp32_path_vs_oracle(n_agents):
    from oracle.spec_fp32 import OracleFp32
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.engine_fp32 import DeployedFp32Model
    model = build_plugin("tiny")
    eng = deploy(model)                                         # a plain model lands on the fp32 path
    assert isinstance(eng, DeployedFp32Model)
    _compare(OracleFp32(eng.state), eng, scene_np(n_agents))
    # graph replay of the fp32 frame
    from quantv2x_amd import synth
    dd = synth.scene_to_torch(scene_np(n_agents), "cuda")
    eager = {k: v.clone() for k, v in eng(dd).items()}
    out = eng.capture(dd)()
    torch.cuda.synchronize()
    for k in eager:
        assert torch.equal(eager[k], out[k]), k


def test_fp32_no_codebook_single_class():
    from oracle.spec_fp32 import OracleFp32
    from quantv2x_amd.engine import deploy
    model = build_plugin("tiny", multiclass=False, codebook=False)
    eng = deploy(model)
    from quantv2x_amd import synth
    sc = scene_np(2)
    want = OracleFp32(eng.state).forward(sc)
    got = eng(synth.scene_to_torch(sc, "cuda"))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got["preds_tensor"].cpu().numpy(), want["preds_tensor"], rtol=2e-4, atol=2e-4)


def test_v2xreal_fp32_frame_exact_and_timed():
    """configs[1] shape in fp32: exact vs the oracle; prints the fp32-path frame time next to the W8A8 one (the reference's claim:
    'W8A8 quantization for substantial speedup')."""
    import copy, time
    from oracle.spec_fp32 import OracleFp32
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.plugin.tools import train_utils
    model = train_utils.create_model(copy.deepcopy(synth.make_hypes("v2xreal"))).eval()
    synth.load_state_dict_numpy(model, synth.make_state_dict(model.state_dict(), seed=1))
    eng = deploy(model)
    sc = synth.make_scene("v2xreal", n_agents=1, seed=3, n_points=60000)
    _compare(OracleFp32(eng.state), eng, sc)
    dd = synth.scene_to_torch(sc, "cuda")
    rep = eng.capture(dd)
    for _ in range(3):
        rep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        rep()
    torch.cuda.synchronize()
    print("fp32 HIP path: %.3f ms per V2X-Real frame" % ((time.perf_counter() - t0) / 20 * 1e3))
