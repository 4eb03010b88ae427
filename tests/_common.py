"""Shared builders for the test-suite: the same seeds ``tests/golden/make_golden.py`` used."""
import copy

import numpy as np
import torch

from quantv2x_amd import synth
from quantv2x_amd.plugin.tools import train_utils

SEED_W, SEED_SCENE, N_POINTS = 1, 3, 3000


def build_plugin(shape="tiny", **kw):
    hy = synth.make_hypes(shape, **kw)
    model = train_utils.create_model(copy.deepcopy(hy)).eval()
    synth.load_state_dict_numpy(model, synth.make_state_dict(model.state_dict(), seed=SEED_W))
    return model


def scene_np(n_agents, shape="tiny", seed=SEED_SCENE, n_points=N_POINTS):
    return synth.make_scene(shape, n_agents=n_agents, seed=seed, n_points=n_points)


def scene(n_agents, shape="tiny", device="cpu", **kw):
    return synth.scene_to_torch(scene_np(n_agents, shape, **kw), device)


def quant_wrap(model, method="minmax"):
    from quantv2x_amd.plugin.tools.inference_quant import wrap
    return wrap(model, method)


def act_quantizers(qt):
    from quantv2x_amd.plugin.tools.inference_quant import activation_quantizers
    return activation_quantizers(qt)


def calibrated_plugin(shape="tiny", n_agents=2, n_points=N_POINTS, **kw):
    """W8A8 min-max ``QuantModel`` frozen after one EMA pass -- the golden ``tiny_w8a8`` recipe."""
    from quantv2x_amd.plugin.tools.inference_quant import calibrate_minmax
    return calibrate_minmax(quant_wrap(build_plugin(shape, **kw)), [scene(n_agents, shape, n_points=n_points)])


def hard_forward(model, dd, taps=None):
    """Forward with the deterministic codebook pair (encode -> decode), stage taps recorded."""
    from quantv2x_amd.plugin.utils.transformation_utils import normalize_pairwise_tfm
    taps = {} if taps is None else taps
    affine = normalize_pairwise_tfm(dd['pairwise_t_matrix'].clone(), model.H, model.W, model.fake_voxel_size)
    f = model.encoder_m1(dd, 'm1'); taps['spatial_features'] = f
    f = model.backbone_m1(f); taps['backbone'] = f
    f = model.shrinker_m1(f); taps['shrinker'] = f
    n, c, h, w = f.shape
    rows = f.permute(0, 2, 3, 1).contiguous().view(-1, c)
    codes = model.codebook.encode(rows)
    dec = model.codebook.decode(codes)
    taps['codes'] = torch.stack([cd[:, 0] for cd in codes]).view(3, n, h, w)
    taps['decoded'] = dec.view(n, h, w, c).permute(0, 3, 1, 2).contiguous()
    fused = model.fusion_net(taps['decoded'], dd['record_len'], affine); taps['fused'] = fused
    taps['preds_tensor'] = torch.cat([model.cls_head(fused), model.reg_head(fused), model.dir_head(fused)], dim=1)
    taps['affine'] = affine
    return taps['preds_tensor']


def sub8(t):
    a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
    return a[:, ::8]
