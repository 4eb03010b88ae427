"""Builders for the SECOND encoder tests."""
import numpy as np
import torch
import torch.nn as nn

from quantv2x_amd import synth

SEED_W, SEED_SCENE = 1, 3


class EncoderOnly(nn.Module):
    """What ``QuantModel`` needs to see: a model whose child ``encoder_m1`` is the SECOND encoder."""

    def __init__(self, enc):
        super().__init__()
        self.encoder_m1 = enc

    def forward(self, data_dict):
        return self.encoder_m1(data_dict, "m1")


def second_scene_np(shape="second_tiny", agents=2, n_points=4000, seed=SEED_SCENE):
    return synth.make_second_scene(shape, agents, seed=seed, n_points=n_points)


def build_second(shape="second_tiny", num_features_out=128):
    from quantv2x_amd.plugin.models.heter_encoders import SECOND
    enc = SECOND(synth.make_second_args(shape, num_features_out)).eval()
    synth.load_state_dict_numpy(enc, synth.make_state_dict(enc.state_dict(), seed=SEED_W))
    return enc


def calibrated_second(shape="second_tiny", agents=2, n_points=4000, num_features_out=128):
    """W8A8 min-max ``QuantModel`` around the encoder, ranges from one pass over the test scene, frozen."""
    from quantv2x_amd.plugin.quant import QuantModel, set_act_quantize_params, set_weight_quantize_params
    wq = dict(n_bits=8, channel_wise=True, scale_method="minmax")
    aq = dict(n_bits=8, channel_wise=False, scale_method="minmax", leaf_param=True)
    qm = QuantModel(EncoderOnly(build_second(shape, num_features_out)), wq, aq).eval()
    sc = second_scene_np(shape, agents, n_points)
    dd = {"inputs_m1": {k: torch.from_numpy(v) for k, v in sc.items()}}
    set_weight_quantize_params(qm)
    set_act_quantize_params(qm, [dd])
    qm.set_quant_state(True, True)
    return qm


def second_model_scene_np(n_agents=2, shape="tiny", n_points=4000, seed=SEED_SCENE):
    return synth.make_scene(shape, n_agents=n_agents, seed=seed, n_points=n_points, encoder="second")


def calibrated_second_model(shape="tiny", n_agents=2, n_points=4000, **kw):
    """The whole collaborative model with a ``core_method: second`` modality, W8A8 min-max, frozen after one pass."""
    import copy
    from quantv2x_amd.plugin.tools import train_utils
    from quantv2x_amd.plugin.tools.inference_quant import calibrate_minmax, wrap
    hy = synth.make_hypes(shape, encoder="second", **kw)
    model = train_utils.create_model(copy.deepcopy(hy)).eval()
    synth.load_state_dict_numpy(model, synth.make_state_dict(model.state_dict(), seed=SEED_W))
    sc = synth.scene_to_torch(second_model_scene_np(n_agents, shape, n_points))
    return calibrate_minmax(wrap(model, "minmax"), [sc])
