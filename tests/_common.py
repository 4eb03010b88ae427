"""Shared builders for the test-suite: the same seeds ``tests/golden/make_golden.py`` used."""
import copy

import numpy as np
import torch

from quantv2x_amd import synth
from quantv2x_amd.plugin.tools import train_utils

SEED_W, SEED_SCENE, N_POINTS = 1, 3, 3000


def build_plugin(shape="tiny", contractive=False, **kw):
    hy = synth.make_hypes(shape, **kw)
    model = train_utils.create_model(copy.deepcopy(hy)).eval()
    make = synth.make_state_dict_contractive if contractive else synth.make_state_dict
    synth.load_state_dict_numpy(model, make(model.state_dict(), seed=SEED_W))
    return model


def build_pyramid_plugin(shape="tiny", **kw):
    hy = synth.make_pyramid_hypes(shape, **kw)
    model = train_utils.create_model(copy.deepcopy(hy)).eval()
    synth.load_state_dict_numpy(model, synth.make_state_dict(model.state_dict(), seed=SEED_W))
    return model


def calibrated_pyramid_plugin(shape="tiny", n_agents=2, n_points=N_POINTS, **kw):
    """The ``pyramid_tiny.npz`` recipe: W8A8 min-max, one EMA pass through the hard (encode -> decode) path, frozen.
    ``codebook=False``: ``heter_pyramid_collab_mc`` (LiDAROnly/lidar_pyramid.yaml), the same network without the compressor."""
    from quantv2x_amd.plugin.tools.inference_quant import calibrate_minmax
    return calibrate_minmax(quant_wrap(build_pyramid_plugin(shape, **kw)), [scene(n_agents, shape, n_points=n_points)])


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


# ---- GPU vs oracle, whole frame (used by the -m gpu suites) ---------------------------------------------------------
FUSE_TOL = dict(rtol=2e-5, atol=2e-5)      # fp32 re-association in the 256-wide dot products + expf


def interior_u8(t):
    """padded i8 BEV [N, H+2, W+2, C] (device) -> uint8 codes [N, H, W, C] (numpy)"""
    return (t[:, 1:-1, 1:-1, :].to(torch.int16) + 128).to(torch.uint8).cpu().numpy()


def head_lsb(state, suffix=""):
    return max(float(state[k + suffix + "/a_delta"]) for k in ("cls_head", "reg_head", "dir_head"))


def compare_frame(orc, eng, sc_np, state, every_layer=True, preds_exact_tol=None, flips_floor=0):
    """One frame through the CPU oracle and through the HIP engine (C ABI): every uint8 activation and every codebook
    index bit-exact, the fused fp32 map within FUSE_TOL, predictions equal up to rare +-1 LSB flips of the head
    quantizer (or within ``preds_exact_tol`` when the head output quantizer is disabled)."""
    otaps, gtaps = {}, {}
    want = orc.forward(sc_np, otaps)
    got = eng(synth.scene_to_torch(sc_np, "cuda"), gtaps)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(interior_u8(gtaps["canvas"]), otaps["canvas"], err_msg="canvas")
    checked = 0
    for name, arr in otaps.items():
        if (name.startswith("backbone_m1.blocks") or name.startswith("shrinker_m1")) and not name.endswith("_q") and name in gtaps:
            if every_layer or name.startswith("shrinker_m1"):
                np.testing.assert_array_equal(interior_u8(gtaps[name]), arr, err_msg=name)
                checked += 1
    assert checked >= 2
    c0 = 0
    for lvl in range(len(eng.deblocks)):
        name = f"backbone_m1.deblocks.{lvl}.0"
        c = otaps[name].shape[-1]
        np.testing.assert_array_equal(interior_u8(gtaps["cat"])[..., c0:c0 + c], otaps[name], err_msg=name)
        c0 += c
    if eng.has_codebook:
        np.testing.assert_array_equal(gtaps["codes"].cpu().numpy().reshape(otaps["codes"].shape), otaps["codes"], err_msg="codebook indices")
    h, w = otaps["fused"].shape[1:3]
    np.testing.assert_allclose(gtaps["fused"].cpu().numpy().reshape(-1, h, w, 256), otaps["fused"], **FUSE_TOL)
    keys = ["cls_preds", "reg_preds", "dir_preds", "preds_tensor"]
    if "cls_preds_single" in want:
        keys += ["cls_preds_single", "reg_preds_single", "dir_preds_single"]
    for key in keys:
        d = np.abs(got[key].cpu().numpy() - want[key])
        if preds_exact_tol is not None:
            assert d.max() <= preds_exact_tol, (key, d.max())
        else:
            lsb = head_lsb(state, "_single" if key.endswith("_single") else "")
            # (``flips_floor``: the rate rule on a tensor of a few thousand elements allows two flips; a caller at tiny shape may grant a count)
            assert d.max() <= lsb * 1.001 and ((d > 1e-5).mean() < 1e-3 or int((d > 1e-5).sum()) <= flips_floor), (key, d.max(), (d > 1e-5).mean())
    return otaps, gtaps, want, got


# ---- heterogeneous scenes (heter_model_baseline.py:169-216): the tiny_heter_w8a8.npz recipe ----------------------------------
HETER_MODALITIES = ["m1", "m2", "m1"]


def heter_scene_np(modalities=HETER_MODALITIES, shape="tiny", seed=SEED_SCENE, n_points=N_POINTS):
    return synth.make_scene(shape, n_agents=len(modalities), seed=seed, n_points=n_points, modalities=modalities)


def calibrated_heter_plugin(shape="tiny", n_points=N_POINTS):
    """Two LiDAR modalities with their own weights, W8A8 min-max, one EMA pass on the [m1, m2, m1] scene, frozen."""
    from quantv2x_amd.plugin.tools.inference_quant import calibrate_minmax
    model = build_plugin(shape, modalities=("m1", "m2"))
    return calibrate_minmax(quant_wrap(model), [synth.scene_to_torch(heter_scene_np(shape=shape, n_points=n_points))])


def hard_forward_heter(model, dd, taps=None):
    """The mirror's heterogeneous forward with the deterministic codebook pair."""
    from quantv2x_amd.plugin.utils.transformation_utils import normalize_pairwise_tfm
    taps = {} if taps is None else taps
    affine = normalize_pairwise_tfm(dd['pairwise_t_matrix'].clone(), model.H, model.W, model.fake_voxel_size)
    per = {}
    for m in model.modality_name_list:
        if m in dd['agent_modality_list']:
            per[m] = getattr(model, 'shrinker_' + m)(getattr(model, 'backbone_' + m)(getattr(model, 'encoder_' + m)(dd, m)))
            taps['shrinker_' + m] = per[m]
    taken = {m: 0 for m in per}
    rows = []
    for m in dd['agent_modality_list']:
        rows.append(per[m][taken[m]])
        taken[m] += 1
    f = torch.stack(rows)
    n, c, h, w = f.shape
    codes = model.codebook.encode(f.permute(0, 2, 3, 1).contiguous().view(-1, c))
    dec = model.codebook.decode(codes)
    taps['codes'] = torch.stack([cd[:, 0] for cd in codes]).view(3, n, h, w)
    taps['decoded'] = dec.view(n, h, w, c).permute(0, 3, 1, 2).contiguous()
    taps['fused'] = model.fusion_net(taps['decoded'], dd['record_len'], affine)
    taps['preds_tensor'] = torch.cat([model.cls_head(taps['fused']), model.reg_head(taps['fused']), model.dir_head(taps['fused'])], dim=1)
    return taps['preds_tensor']


def heter_oracle_forward(states, sc_np, taps=None):
    """The CPU oracle on a heterogeneous scene: every modality's own oracle (its PTQ state) runs a1-a6 on that modality's agents, the
    code planes are put in agent order, the ego modality's oracle decodes, fuses and runs the heads."""
    from oracle.spec import Oracle
    taps = {} if taps is None else taps
    agents = list(sc_np["agent_modality_list"])
    per = {}
    for m, st in states.items():
        idx = [i for i, a in enumerate(agents) if a == m]
        if not idx:
            continue
        orc = Oracle(st)
        sub = {"inputs_m1": sc_np["inputs_" + m], "agent_modality_list": ["m1"] * len(idx)}
        canvas, cq = orc.canvas(sub, len(idx))                             # PointPillar or SECOND, as this modality's state says
        mt = {}
        cat, cat_q = orc.backbone(canvas, cq, mt)
        shr, shr_q = orc.shrinker(cat, cat_q, mt)
        per[m] = (idx, orc.encode(shr, shr_q).reshape(-1, len(idx), shr.shape[1] * shr.shape[2]), shr)
        taps["shrinker_" + m] = shr
    main = Oracle(states[agents[0]] if agents[0] in states else next(iter(states.values())))
    lv, hw = next(iter(per.values()))[1].shape[0], next(iter(per.values()))[1].shape[2]
    codes = np.zeros((lv, len(agents), hw), np.uint8)
    for m, (idx, c, _) in per.items():
        codes[:, idx, :] = c
    shr = next(iter(per.values()))[2]
    h, w = shr.shape[1], shr.shape[2]
    taps["codes"] = codes.reshape(lv, len(agents), h, w)
    feats = main.decode(codes.reshape(lv, -1)).reshape(len(agents), h, w, 256)
    fused = main.fuse(feats, sc_np["pairwise_t_matrix"], sc_np["record_len"])
    taps["fused"] = fused
    cls, reg, dr = main.heads(fused)
    out = {"cls_preds": cls, "reg_preds": reg, "dir_preds": dr, "preds_tensor": np.concatenate([cls, reg, dr], axis=1)}
    if bool(main.s["meta/supervise_single"]):
        s_cls, s_reg, s_dir = main.heads(feats, "_single")
        out.update({"cls_preds_single": s_cls, "reg_preds_single": s_reg, "dir_preds_single": s_dir})
    return out


# ---- MIXED-encoder scenes (heter_model_baseline.py:47-59: the encoder class is picked per modality): m1 PointPillar + m3 SECOND -------
MIXED_MODALITIES = ["m1", "m3", "m1"]
MIXED_ENCODERS = {"m3": "second"}


def mixed_scene_np(modalities=MIXED_MODALITIES, shape="tiny", seed=SEED_SCENE, n_points=N_POINTS, **kw):
    return synth.make_scene(shape, n_agents=len(modalities), seed=seed, n_points=n_points, modalities=modalities, encoders=MIXED_ENCODERS, **kw)


def calibrated_mixed_plugin(shape="tiny", n_points=N_POINTS, **kw):
    """One PointPillar modality (m1) and one SECOND modality (m3) under the shared codebook / fusion / heads, W8A8 min-max, one EMA pass on
    the [m1, m3, m1] scene, frozen."""
    from quantv2x_amd.plugin.tools.inference_quant import calibrate_minmax
    model = build_plugin(shape, modalities=("m1", "m3"), encoders=MIXED_ENCODERS, **kw)
    return calibrate_minmax(quant_wrap(model), [synth.scene_to_torch(mixed_scene_np(shape=shape, n_points=n_points))])


# ---- a scene seen from agent r (the every-rank-is-its-own-ego mode of the N-GPU path, dist.py) ----------------------------------------
def ego_first(n_agents, ego):
    """The reference has no ``ego`` argument: the ego is agent 0 of the scene and only its row is kept (fusion_in_one.py:141-147,
    ``i = 0 # ego``).  The scene as agent ``ego`` sees it is therefore the same agents presented ego-first."""
    return [ego] + [a for a in range(n_agents) if a != ego]


def ego_view(feats, pairwise_b, n_agents, ego):
    """(feats [n, ...] re-ordered ego-first, pairwise [1, L, L, 4, 4] with rows and columns permuted the same way: T'[i][j] =
    T[order[i]][order[j]], the matrix get_pairwise_transformation builds for that agent order, transformation_utils.py:21-66)."""
    order = ego_first(n_agents, ego)
    t = np.array(pairwise_b, dtype=np.float64, copy=True)
    sub = t[:n_agents, :n_agents][order][:, order]
    t[:n_agents, :n_agents] = sub
    return feats[order], t[None]
