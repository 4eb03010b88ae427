"""W4A8 -- the configuration the reference's own PTQ script runs (scripts/inference/inference_quant.sh:1 ``--n_bits_w 4 --n_bits_a 8``,
opencood/tools/inference_quant.py:102-104, 225-232; ``set_first_last_layer_to_8bit`` quant_model.py:115-127; ``bitwidth_refactor``
quant_layer.py:337-340).  A 4-bit code and its zero point lie in [0, 15], stay uint8 and run on the same int8 kernels: the mirror gives
the reference's quantizers (``tests/golden/tiny_w4a8.npz``), ``export_ptq_state`` freezes them, the CPU oracle reproduces the
reference's fake-quant codes layer by layer, and (``-m gpu``) the deployed path equals the oracle bit for bit, tiny and V2X-Real."""
import numpy as np
import pytest
import torch

from _common import build_plugin, scene, scene_np

from quantv2x_amd.ptq_state import export_ptq_state


def calibrated_w4a8(shape="tiny", n_agents=2, n_points=3000):
    from quantv2x_amd.plugin.tools.inference_quant import calibrate_minmax, wrap
    return calibrate_minmax(wrap(build_plugin(shape), n_bits_w=4, n_bits_a=8, first_last_8bit=True), [scene(n_agents, shape, n_points=n_points)])


@pytest.fixture(scope="module")
def state():
    torch.set_num_threads(1)
    return export_ptq_state(calibrated_w4a8())


def test_mirror_quantizers_are_the_references(golden, state):
    import test_oracle_golden as T
    g = golden["tiny_w4a8"]
    np.testing.assert_array_equal(state["meta/w_bits"], g["bits_w"])
    assert list(g["bits_w"]) == [8] + [4] * (len(g["bits_w"]) - 2) + [8] and (g["bits_a"] == 8).all()
    T.test_exported_state_matches_reference_quantizers({"tiny_w8a8": g}, state)
    for n, b in zip(state["meta/module_names"], state["meta/w_bits"]):
        assert int(state[f"{n}/w_code"].max()) <= 2 ** int(b) - 1 and float(state[f"{n}/w_zp"].max()) <= 2 ** int(b) - 1
        assert int(state[f"{n}/w_code"].max()) == int(g[str(n).replace(".", "/") + "/w_code_max"])


def test_oracle_vs_reference_fake_quant_w4a8(golden, state):
    import test_oracle_golden as T
    T.test_integer_path_vs_reference_fake_quant({"tiny_w8a8": golden["tiny_w4a8"]}, state)


def test_export_refuses_sub_8_bit_activations():
    from quantv2x_amd.plugin.tools.inference_quant import calibrate_minmax, wrap
    torch.set_num_threads(1)
    qt = calibrate_minmax(wrap(build_plugin(), n_bits_w=4, n_bits_a=4), [scene(2)])
    with pytest.raises(NotImplementedError, match="WxA8"):
        export_ptq_state(qt)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,n_agents,n_points", [("tiny", 2, 3000), ("v2xreal", 1, 60000)])
def test_deployed_w4a8_equals_oracle(shape, n_agents, n_points, state):
    from _common import compare_frame
    from oracle.spec import Oracle
    from quantv2x_amd.engine import deploy
    torch.set_num_threads(8)
    st = state if shape == "tiny" else export_ptq_state(calibrated_w4a8(shape, n_agents, n_points))
    assert int(st["meta/w_bits"].min()) == 4
    compare_frame(Oracle(st), deploy(state=st), scene_np(n_agents, shape, n_points=n_points), st, every_layer=(shape == "tiny"))
