"""Deploy inputs other than the min-max recipe, on the GPU against the oracle: an AdaRound hard-mask
state, heads with the output quantizer disabled, an MSE-calibrated state.  Tiny shape; same bar as test_hip_parity."""
import numpy as np
import pytest
import torch

from _common import compare_frame, scene_np
from _states import adaround_plugin, mse_plugin, output_quant_off_plugin

pytestmark = pytest.mark.gpu
torch.set_num_threads(8)


def _deploy(qt):
    from oracle.spec import Oracle
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(qt)
    return state, Oracle(state), deploy(state=state)


@pytest.mark.parametrize("n_agents", [1, 3])
def test_adaround_hard_mask_state(n_agents):
    state, orc, eng = _deploy(adaround_plugin())
    compare_frame(orc, eng, scene_np(n_agents), state)


def test_heads_without_output_quantizer():
    """a_off: the heads return the fp32 accumulator (+ bias); no LSB to flip, so the bound is the fp32 one"""
    state, orc, eng = _deploy(output_quant_off_plugin())
    assert all(bool(state[h + "/a_off"]) for h in ("cls_head", "reg_head", "dir_head"))
    _, _, want, got = compare_frame(orc, eng, scene_np(2), state, preds_exact_tol=2e-4)
    # and they are not on a quantization grid any more
    d = float(state["cls_head_single/a_delta"])
    frac = np.abs(got["cls_preds"].cpu().numpy() / d - np.round(got["cls_preds"].cpu().numpy() / d))
    assert (frac > 1e-3).mean() > 0.5


@pytest.mark.parametrize("n_agents", [2])
def test_mse_calibrated_state(n_agents):
    state, orc, eng = _deploy(mse_plugin())
    compare_frame(orc, eng, scene_np(n_agents), state)


def test_reconstruction_on_the_gpu_then_deploy():
    """SURVEY §8(f) rank 4 end to end: the AdaRound / QDrop reconstruction loops run on the MI355X in PyTorch-ROCm (autograd),
    the frozen result (hard rounding masks, learned activation step sizes) deploys on the HIP int8 path, bit-exact vs the oracle."""
    from _common import build_plugin, scene
    from quantv2x_amd.plugin.quant import AdaRoundQuantizer, QuantModule
    from quantv2x_amd.plugin.tools import inference_quant as IQ
    fp, qt = IQ.wrap_pair(build_plugin("tiny"))
    fp.cuda(); qt.cuda()
    cali = [scene(2, seed=3 + i, device="cuda") for i in range(3)]
    seen = []
    IQ.recon_model(qt, fp, IQ.recon_kwargs(cali, iters_w=25, dc_iters=3, verbose=False, seed=0), log=seen.append)
    assert len(seen) == 9
    for m in qt.modules():
        if isinstance(m, QuantModule):
            assert isinstance(m.weight_quantizer, AdaRoundQuantizer) and m.weight_quantizer.alpha.is_cuda and m.trained
    state, orc, eng = _deploy(qt)
    for n_agents in (1, 2):
        compare_frame(orc, eng, scene_np(n_agents), state)


def test_pyramid_reconstruction_on_the_gpu_then_deploy():
    """The same for the HEAL Pyramid model: PFN -> agent ResNet block -> ``QuantPyramidFusion`` as one unit (``pyramid_reconstruction``)
    -> shrink_conv -> heads reconstructed on the GPU, exported, deployed on the Pyramid engine, every stage bit-exact vs its oracle."""
    from _common import build_pyramid_plugin, scene
    from test_hip_pyramid import compare_pyramid_frame
    from oracle.spec_pyramid import OraclePyramid
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.plugin.quant import AdaRoundQuantizer, QuantModule
    from quantv2x_amd.plugin.tools import inference_quant as IQ
    from quantv2x_amd.ptq_state import export_ptq_state
    fp, qt = IQ.wrap_pair(build_pyramid_plugin("tiny"))
    fp.cuda(); qt.cuda()
    cali = [scene(2, seed=3 + i, device="cuda") for i in range(2)]
    seen = []
    IQ.recon_model(qt, fp, IQ.recon_kwargs(cali, iters_w=10, dc_iters=2, verbose=False, seed=0), log=seen.append)
    assert [s.split()[-1] for s in seen] == ["0", "backbone_m1", "pyramid_backbone", "shrink_conv", "cls_head", "reg_head", "dir_head"]
    assert all(isinstance(m.weight_quantizer, AdaRoundQuantizer) and m.weight_quantizer.alpha.is_cuda for m in qt.modules() if isinstance(m, QuantModule))
    st = export_ptq_state(qt)
    eng = deploy(state=st)
    compare_pyramid_frame(OraclePyramid(st), eng, scene_np(2), st)
