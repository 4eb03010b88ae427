"""Full-size (BASELINE.json shapes) checks on the GPU.  The C/OpenMP oracle finishes a V2X-Real agent-frame in well under
a second on the GPU box's host cores, so every BASELINE config is compared EXACTLY end to end (uint8 activations and
codebook indices bit-exact, fused map 2e-5, predictions up to rare +-1 LSB head-quantizer flips):

  configs[1]  V2X-Real, 1 agent                  test_v2xreal_frames_exact[1]
  configs[2]  V2X-Real, 2 agents (line layout)   test_v2xreal_frames_exact[2]
  configs[3]  V2X-Real, 4 agents (ring layout)   test_v2xreal_frames_exact[4]
  configs[4]  OPV2V 512 x 512 grid, 1 and 8 agents (max_cav raised to 8), single-class heads   test_opv2v_*

plus size-independent properties at the 4-agent shape."""
import os

import numpy as np
import pytest
import torch

from _common import compare_frame, head_lsb

pytestmark = pytest.mark.gpu


def _calibrated(shape, **kw):
    import copy
    from quantv2x_amd import synth
    from quantv2x_amd.plugin.tools import inference_quant, train_utils
    from quantv2x_amd.ptq_state import export_ptq_state
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    model = train_utils.create_model(copy.deepcopy(synth.make_hypes(shape, **kw))).eval()
    synth.load_state_dict_numpy(model, synth.make_state_dict(model.state_dict(), seed=1))
    calib = synth.scene_to_torch(synth.make_scene(shape, n_agents=1, seed=3, n_points=60000))
    return export_ptq_state(inference_quant.calibrate_minmax(inference_quant.wrap(model), [calib]))


@pytest.fixture(scope="module")
def v2xreal():
    from oracle.spec import Oracle
    from quantv2x_amd.engine import deploy
    state = _calibrated("v2xreal")
    return state, deploy(state=state), Oracle(state)


@pytest.fixture(scope="module")
def opv2v():
    from oracle.spec import Oracle
    from quantv2x_amd.engine import deploy
    state = _calibrated("opv2v", multiclass=False)
    return state, deploy(state=state), Oracle(state)


@pytest.mark.parametrize("n_agents,layout,seed", [(1, "line", 3), (2, "line", 3), (4, "ring", 5)])
def test_v2xreal_frames_exact(v2xreal, n_agents, layout, seed):
    """BASELINE.json configs[1..3]: 60k points per agent, 704 x 200 grid, 3 x 35 200 indices per agent."""
    from quantv2x_amd import synth
    state, eng, orc = v2xreal
    sc = synth.make_scene("v2xreal", n_agents=n_agents, seed=seed, n_points=60000, layout=layout)
    compare_frame(orc, eng, sc, state)


def test_opv2v_single_agent_frame_exact(opv2v):
    """configs[4] shape, one agent: 256 x 256 maps (ragged 5 x 32 wide-conv patches, 65 536-row encode)."""
    from quantv2x_amd import synth
    state, eng, orc = opv2v
    sc = synth.make_scene("opv2v", n_agents=1, seed=3, n_points=60000)
    _, gt, _, got = compare_frame(orc, eng, sc, state)
    assert gt["codes"].shape[-1] == 65536 and got["preds_tensor"].shape == (1, 20, 256, 256)


def test_opv2v_eight_agents_exact(opv2v):
    """configs[4]: 8-agent dense scene (ring layout, max_cav 8), single-class heads -- the whole frame against the oracle."""
    from quantv2x_amd import synth
    state, eng, orc = opv2v
    sc = synth.make_scene("opv2v", n_agents=8, seed=7, n_points=60000, layout="ring")
    _, gt, _, got = compare_frame(orc, eng, sc, state, every_layer=False)
    assert got["preds_tensor"].shape == (1, 20, 256, 256) and got["cls_preds_single"].shape == (8, 2, 256, 256)
    assert gt["codes"].shape == (3, 8, 65536)


def test_v2xreal_every_agent_as_ego(v2xreal):
    """configs[3] (4 agents, ring layout) seen from EVERY agent: the N-GPU path's default mode (``ego = rank``) against the oracle on the
    scene presented ego-first (fusion_in_one.py:131-151 has no ego argument: agent 0 is the ego) -- every a7-a11 entry point, the
    gathered wire layout with device-built pairwise matrices included (tests/test_hip_ego.py)."""
    from quantv2x_amd import synth
    from test_hip_ego import every_entry_point_as_ego
    state, eng, orc = v2xreal
    sc = synth.make_scene("v2xreal", n_agents=4, seed=5, n_points=60000, layout="ring")
    worst = every_entry_point_as_ego(state, eng, orc, sc, 4, range(4), layout="ring")
    assert worst < 2e-4


def test_v2xreal_properties_four_agents(v2xreal):
    """configs[3] shape (4 agents, ring layout): agent permutation-equivariance of the encode, determinism, identity-pose
    fusion of identical agents returns the agent itself, and the N = 1 slice of a batch equals the single-agent run."""
    from quantv2x_amd import synth
    state, eng, _ = v2xreal
    sc = synth.make_scene("v2xreal", n_agents=4, seed=5, n_points=60000, layout="ring")
    dd = synth.scene_to_torch(sc, "cuda")
    t1, t2 = {}, {}
    a = {k: v.clone() for k, v in eng(dd, t1).items()}
    codes1 = t1["codes"].clone()
    b = eng(dd, t2)
    for k in a:
        assert torch.equal(a[k], b[k]), k                                  # deterministic
    # encode is per agent: agent 2 alone gives the same code planes as agent 2 inside the batch
    co = dd["inputs_m1"]["voxel_coords"]
    sel = co[:, 0] == 2
    solo = {"voxel_features": dd["inputs_m1"]["voxel_features"][sel].contiguous(),
            "voxel_coords": co[sel].clone().contiguous(), "voxel_num_points": dd["inputs_m1"]["voxel_num_points"][sel].contiguous()}
    solo["voxel_coords"][:, 0] = 0
    codes_solo = eng.encode_agents(solo, 1).clone()
    assert torch.equal(codes_solo[:, 0], codes1[:, 2])
    # identical agents at identical poses: attention over equal rows returns the row (softmax weights 1/N each)
    hw = eng.fh * eng.fw
    rep = codes_solo.expand(-1, 3, -1).contiguous()                          # [levels, 3, hw]
    eye = torch.eye(4, dtype=torch.float64, device="cuda").expand(5, 5, 4, 4).contiguous()
    one = eng.fuse_and_heads(codes_solo.contiguous(), hw, hw, eye, 1)["preds_tensor"].clone()
    three = eng.fuse_and_heads(rep, hw, 3 * hw, eye, 3)["preds_tensor"]
    d = (one - three).abs()
    assert float(d.max()) <= head_lsb(state) * 1.001 and float((d > 1e-5).float().mean()) < 1e-3


def test_v2xreal_batch_of_eight_frames_exact(v2xreal):
    """Eight single-agent frames in one batch (rounds 1-2's throughput configuration; the bench's own batch of 32 is
    test_bench_configuration_batch32_two_streams_exact below).  At this size every stride-1 layer of the
    backbone runs on the halo-patch kernel (64 / 128 output channels per workgroup, the 25 x 88 level split in two channel halves);
    every uint8 activation of the encode side and every codebook index against the oracle, frame by frame."""
    from quantv2x_amd import synth
    state, eng, orc = v2xreal
    scenes = [synth.make_scene("v2xreal", n_agents=1, seed=11 + f, n_points=60000) for f in range(8)]
    parts = []
    for f, sc in enumerate(scenes):
        part = {k: v.copy() for k, v in sc["inputs_m1"].items()}
        part["voxel_coords"][:, 0] += f
        parts.append(part)
    inputs = {k: torch.from_numpy(np.concatenate([p[k] for p in parts])).cuda() for k in parts[0]}
    taps = {}
    codes = eng.encode_agents(inputs, 8, taps).cpu().numpy()            # [levels, 8, H*W]
    torch.cuda.synchronize()
    names = [f"backbone_m1.blocks.{l}.{i + 1}" for l, n in enumerate(eng.layer_nums) for i in range(n + 1)] + \
            ["shrinker_m1.layers.0.double_conv.0", "shrinker_m1.layers.0.double_conv.1"]
    from _common import interior_u8
    got = {n: interior_u8(taps[n]) for n in names}
    for f, sc in enumerate(scenes):
        ot = {}
        orc.forward(sc, ot)
        for n in names:
            np.testing.assert_array_equal(got[n][f], ot[n][0], err_msg=f"frame {f}: {n}")
        np.testing.assert_array_equal(codes[:, f].reshape(ot["codes"].shape[0], -1), ot["codes"].reshape(ot["codes"].shape[0], -1), err_msg=f"frame {f}: codes")


def test_bench_configuration_batch32_two_streams_exact():
    """bench.py's own configuration, built by bench.py's own functions: B = 32 single-agent V2X-Real frames per HIP
    graph (``bench.build_engine`` + ``bench.frame_batch`` + ``eng.capture(full)``), TWO engines on TWO streams, replayed twice in the
    bench's round-robin order.  288 MB canvases, 2-9 persistent items per conv workgroup, 9 M-thread grids: the codebook indices of all
    32 frames of BOTH engines and the shrinker's uint8 map of 4 frames, bit for bit against the oracle; the predictions of the
    graphs' output dict against it for the same 4 frames."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from _common import head_lsb, interior_u8
    from oracle.spec import Oracle
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    B, F = 32, 2
    wl = bench.workload_for(1)
    bench.SHAPE, bench.N_POINTS = wl["shape"], wl["n_points"]
    state, eng, _, _ = bench.build_engine(min(32, os.cpu_count() or 8), multiclass=wl["multiclass"])
    _, full, _, _ = bench.frame_batch(1, 0, B, torch.device("cuda"), layout=wl["layout"], max_cav=wl["max_cav"])
    engines = [eng] + [deploy(state=state) for _ in range(F - 1)]
    streams = [torch.cuda.Stream() for _ in range(F)]
    reps = []
    for e, st in zip(engines, streams):
        with torch.cuda.stream(st):
            reps.append(e.capture(full))
    torch.cuda.synchronize()
    outs = [None] * F
    for i in range(2 * F):                               # bench.py's step(): graph i % F on stream i % F, nothing waits in between
        with torch.cuda.stream(streams[i % F]):
            outs[i % F] = reps[i % F]()
    torch.cuda.synchronize()
    orc = Oracle(state)
    codes = [e._workspace(B)["codes"].cpu().numpy() for e in engines]               # [levels, B, H*W]
    preds = [o["preds_tensor"].cpu().numpy() for o in outs]
    lsb = head_lsb(state)
    assert torch.equal(engines[0]._workspace(B)["s1"], engines[1]._workspace(B)["s1"])
    for f in range(B):
        sc = synth.make_scene(wl["shape"], n_agents=1, seed=3 + f, n_points=wl["n_points"], layout=wl["layout"], max_cav=wl["max_cav"])
        ot = {}
        want = orc.forward(sc, ot)
        oc = ot["codes"].reshape(ot["codes"].shape[0], -1)
        for k in range(F):
            np.testing.assert_array_equal(codes[k][:, f], oc, err_msg=f"engine {k} frame {f}: codebook indices")
        if f % 8 == 0:
            for k in range(F):
                got = interior_u8(engines[k]._workspace(B)["s1"][f:f + 1])[0]
                np.testing.assert_array_equal(got, ot["shrinker_m1.layers.0.double_conv.1"][0], err_msg=f"engine {k} frame {f}: shrinker map")
                d = np.abs(preds[k][f] - want["preds_tensor"][0])
                assert d.max() <= lsb * 1.001 and (d > 1e-5).mean() < 1e-3, (k, f, d.max())


def _mixed(shape, modalities, layout, seed, n_points_calib=20000, **kw):
    """A mixed-encoder model (m1 PointPillar + m3 SECOND: heter_model_baseline.py:47-59) calibrated and deployed at full size, one scene
    through it and through the per-modality oracles."""
    import copy
    from _common import MIXED_ENCODERS, heter_oracle_forward
    from quantv2x_amd import synth
    from quantv2x_amd.engine import DeployedHeterModel, deploy
    from quantv2x_amd.plugin.tools import inference_quant, train_utils
    from quantv2x_amd.ptq_state import export_ptq_state
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    model = train_utils.create_model(copy.deepcopy(synth.make_hypes(shape, modalities=("m1", "m3"), encoders=MIXED_ENCODERS, **kw))).eval()
    synth.load_state_dict_numpy(model, synth.make_state_dict(model.state_dict(), seed=1))
    calib = synth.scene_to_torch(synth.make_scene(shape, n_agents=2, seed=3, n_points=n_points_calib, modalities=["m1", "m3"], encoders=MIXED_ENCODERS))
    qt = inference_quant.calibrate_minmax(inference_quant.wrap(model), [calib])
    states = {m: export_ptq_state(qt, modality=m) for m in ("m1", "m3")}
    eng = deploy(qt)
    assert isinstance(eng, DeployedHeterModel)
    sc = synth.make_scene(shape, n_agents=len(modalities), seed=seed, n_points=60000, layout=layout, modalities=modalities, encoders=MIXED_ENCODERS,
                          max_cav=max(5, len(modalities)))
    otaps, gtaps = {}, {}
    want = heter_oracle_forward(states, sc, otaps)
    got = eng(synth.scene_to_torch(sc, "cuda"), gtaps)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(gtaps["codes"].cpu().numpy().reshape(otaps["codes"].shape), otaps["codes"])     # every agent's wire indices
    h, w = otaps["fused"].shape[1:3]
    from _common import FUSE_TOL
    np.testing.assert_allclose(gtaps["fused"].cpu().numpy().reshape(-1, h, w, 256), otaps["fused"], **FUSE_TOL)
    d = np.abs(got["preds_tensor"].cpu().numpy() - want["preds_tensor"])
    assert d.max() <= head_lsb(states["m1"]) * 1.001 and (d > 1e-5).mean() < 1e-3
    return otaps, got


def test_v2xreal_mixed_encoder_scene_exact():
    """One heterogeneous V2X-Real scene: a PointPillar agent and a SECOND agent (0.1 m voxels over the same range) under the shared codebook,
    fusion and heads -- 2 x 3 x 35 200 wire indices against the per-modality oracles, bit for bit."""
    otaps, got = _mixed("v2xreal", ["m1", "m3"], "line", seed=3)
    assert otaps["codes"].shape == (3, 2, 100, 352) and got["preds_tensor"].shape == (1, 72, 100, 352)


def test_opv2v_mixed_encoders_eight_agents_exact():
    """BASELINE configs[4] is OPV2V-H -- H for heterogeneous: eight agents, four PointPillar and four SECOND, on the 512 x 512 grid
    (SECOND: 2048 x 2048 x 40 voxels), max_cav 8, single-class heads."""
    otaps, got = _mixed("opv2v", ["m1", "m3"] * 4, "ring", seed=7, multiclass=False)
    assert otaps["codes"].shape == (3, 8, 256, 256) and got["preds_tensor"].shape == (1, 20, 256, 256) and got["cls_preds_single"].shape[0] == 8
