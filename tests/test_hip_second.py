"""a13 on the GPU: the HIP SECOND encoder against the CPU restatement (oracle/spec_second.py), bit for bit."""
import numpy as np
import pytest
import torch

from _common_second import calibrated_second, second_scene_np

pytestmark = pytest.mark.gpu


def _rows_by_key(codes, coords, shape):
    c = coords.astype(np.int64)
    key = ((c[:, 0] * shape[0] + c[:, 1]) * shape[1] + c[:, 2]) * shape[2] + c[:, 3]
    order = np.argsort(key)
    return key[order], codes[order]


@pytest.mark.parametrize("shape,agents,n_points,cout", [("second_tiny", 2, 4000, 128), ("second_small", 1, 30000, 128), ("second_tiny", 1, 4000, 64)])
def test_second_encoder_matches_the_oracle(shape, agents, n_points, cout):
    from oracle.spec_second import OracleSecond
    from quantv2x_amd.engine_second import DeployedSecondEncoder
    from quantv2x_amd.ptq_state import export_second_state
    from quantv2x_amd import synth
    qm = calibrated_second(shape, agents, n_points, num_features_out=cout)
    state = export_second_state(qm.model.encoder_m1)
    sc = second_scene_np(shape, agents, n_points)
    to, tg = {}, {}
    want = OracleSecond(state).forward(sc, to, batch_size=agents)
    eng = DeployedSecondEncoder(state, "cuda:0", agents=agents, max_voxels=synth.SECOND_SHAPES[shape][2])
    inp = {k: torch.from_numpy(v).cuda() for k, v in sc.items()}
    for rep in range(2):                                              # the second pass proves the index volumes were left clean
        tg.clear()
        bev = eng(inp, tg)
        torch.cuda.synchronize()
        for i in range(int(state["second/n_layers"])):
            oc, oi, osh = to[f"second/{i}"]
            f, coords, count, co, gsh = tg[f"second/{i}"]
            n = int(count.item())
            assert n == len(oi) and list(gsh) == list(osh), f"layer {i}: {n} sites, the oracle has {len(oi)}"
            gk, gc = _rows_by_key((f[:n, :co].to(torch.int16) + 128).cpu().numpy().astype(np.uint8), coords[:n].cpu().numpy(), gsh)
            ok, occ = _rows_by_key(oc, oi, osh)
            assert np.array_equal(gk, ok), f"layer {i}: active sites differ"
            if i >= 2:                                                # levels a SparseConv3d opened: rows in raster order, like the checker's
                assert np.array_equal(coords[:n].cpu().numpy(), oi), f"layer {i}: rows are not in raster order"
            assert np.array_equal(gc, occ), f"layer {i}: {(gc != occ).sum()} codes differ"
        assert np.array_equal(eng.dense_codes(bev).cpu().numpy(), want)
    for l in eng.levels:
        assert int((l.volume != -1).sum().item()) == 0


def test_second_encoder_close_to_the_torch_mirror():
    from quantv2x_amd.engine_second import deploy_second
    qm = calibrated_second("second_tiny", 2, 4000)
    sc = second_scene_np("second_tiny", 2, 4000)
    dd = {"inputs_m1": {k: torch.from_numpy(v) for k, v in sc.items()}}
    with torch.no_grad():
        ref = qm(dd).numpy()
    eng = deploy_second(qm, device="cuda:0", agents=2, max_voxels=4096)
    got = eng.dequant(eng({k: torch.from_numpy(v).cuda() for k, v in sc.items()})).cpu().numpy()
    lsb = eng.out_q[0]
    d = np.abs(got - ref)
    assert d.max() <= 6.001 * lsb and (d > 1e-4).mean() < 2e-2, (d.max() / lsb, (d > 1e-4).mean())     # whole path: see test_second_cpu.py


def test_empty_and_single_voxel():
    from oracle.spec_second import OracleSecond
    from quantv2x_amd.engine_second import DeployedSecondEncoder
    from quantv2x_amd.ptq_state import export_second_state
    qm = calibrated_second("second_tiny", 1, 4000)
    state = export_second_state(qm.model.encoder_m1)
    eng = DeployedSecondEncoder(state, "cuda:0", agents=1, max_voxels=4096)
    zp = int(state["second/11/a_zp"])
    empty = {"voxel_features": torch.zeros((0, 5, 4), device="cuda"), "voxel_coords": torch.zeros((0, 4), dtype=torch.int32, device="cuda"),
             "voxel_num_points": torch.zeros((0,), dtype=torch.int32, device="cuda")}
    assert int((eng.dense_codes(eng(empty)) != zp).sum().item()) == 0
    one = {"voxel_features": np.zeros((1, 5, 4), np.float32), "voxel_coords": np.array([[0, 17, 33, 65]], np.int32), "voxel_num_points": np.array([2], np.int32)}
    one["voxel_features"][0, :2] = [[0.1, -0.2, -1.2, 0.5], [0.12, -0.22, -1.25, 0.7]]
    want = OracleSecond(state).forward(one, batch_size=1)
    got = eng.dense_codes(eng({k: torch.from_numpy(v).cuda() for k, v in one.items()})).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n_agents", [1, 2])
def test_whole_model_with_a_second_modality(n_agents):
    """The SECOND encoder in front of the int8 2-D path: every uint8 activation and codebook index equal to the oracle's."""
    from _common import compare_frame
    from _common_second import calibrated_second_model, second_model_scene_np
    from oracle.spec import Oracle
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    state = export_ptq_state(calibrated_second_model(n_agents=2))
    eng = deploy(state=state, device="cuda:0")
    eng.second_max_voxels = 4096
    sc = second_model_scene_np(n_agents)
    compare_frame(Oracle(state), eng, sc, state)
    compare_frame(Oracle(state), eng, sc, state, every_layer=False)      # again: the index volumes were left clean


def test_second_model_as_a_hip_graph():
    """The whole frame with the SECOND modality captured into one hipGraph (row counts never leave the device): replay == eager."""
    from _common_second import calibrated_second_model, second_model_scene_np
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    eng = deploy(state=export_ptq_state(calibrated_second_model(n_agents=2)), device="cuda:0")
    eng.second_max_voxels = 4096
    sc = synth.scene_to_torch(second_model_scene_np(2), "cuda")
    want = {k: v.clone() for k, v in eng(sc).items()}
    rep = eng.capture(sc)
    for _ in range(2):
        out = rep()
        torch.cuda.synchronize()
        for k in want:
            assert torch.equal(out[k], want[k]), k
