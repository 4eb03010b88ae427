"""Box-level end-to-end agreement: the same scene through

  (i)  the torch mirror of the reference under W8A8 fake-quant (``QuantModel``; codebook on its deterministic encode -> decode
       pair) -> the CPU post-processor (oracle/postprocess.py), and
  (ii) the deployed HIP path -> ``qv2x_postprocess_f32`` behind ``VoxelPostprocessor3Heads.post_process``,

compared as DETECTIONS: boxes of the same class matched at bottom-face IoU >= 0.7 (the strictest threshold of the reference's
evaluation, opencood/utils/eval_utils.py:40-193), and the score difference of the matched pairs.  Layer by layer the two paths
agree to <= 1 LSB on < 5e-4 of the activations (tests/test_oracle_golden.py); end to end the integer path and the fp32-emulated
path drift (a random-weight 21-layer stack amplifies single flips), so this is the available stand-in for "matched detection
output" without checkpoints or datasets.  The numbers are written to gpurun_out/box_agreement_<shape>.json."""
import json
import os

import numpy as np
import pytest
import torch

from _common import calibrated_plugin, scene_np

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _match(ca, la, cb, lb, thr=0.7):
    """for every box of a: best bottom-face IoU with a same-label box of b -> (matched mask, index in b, iou)"""
    from oracle.postprocess import quad_area, quad_intersection_area
    qa, qb = ca[:, :4, :2].astype(np.float64), cb[:, :4, :2].astype(np.float64)
    cena, cenb = qa.mean(1), qb.mean(1)
    areab = [quad_area(q) for q in qb]
    best, arg = np.zeros(len(qa)), np.full(len(qa), -1)
    for i in range(len(qa)):
        near = np.nonzero((np.abs(cenb - cena[i]).max(1) < 6.0) & (lb == la[i]))[0]
        ai = quad_area(qa[i])
        for j in near:
            inter = quad_intersection_area(qa[i], qb[j])
            iou = inter / (ai + areab[j] - inter) if ai + areab[j] - inter > 0 else 0.0
            if iou > best[i]:
                best[i], arg[i] = iou, j
    return best >= thr, arg, best


@pytest.mark.parametrize("model,shape,n_agents,n_points", [("attfuse", "tiny", 2, 3000), ("attfuse", "v2xreal", 2, 60000),
                                                          ("pyramid", "tiny", 2, 3000), ("pyramid", "v2xreal", 2, 60000),
                                                          ("attfuse-contractive", "tiny", 2, 3000), ("attfuse-contractive", "v2xreal", 2, 60000)])
def test_detections_agree_with_the_fake_quant_mirror(model, shape, n_agents, n_points):
    """``*-contractive``: the same network with ``synth.make_state_dict_contractive`` -- convolutions that pass a +-1 code flip on
    instead of multiplying it by ~27 per layer as He-normal weights do.  There the integer path and the fp32-emulated path must give the
    SAME detections (>= 95 % matched): what is left of the disagreement on the random-weight sets is the network's own chaos, for which
    ``fp32_vs_w8a8_mirror_*`` (the reference arithmetic against itself, quantization on / off) is the yardstick."""
    from _common import calibrated_pyramid_plugin
    from oracle import postprocess as P
    from test_postprocess_oracle import MC_CFGS, interleave, mc_params
    from quantv2x_amd import synth
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.plugin.data_utils.post_processor import build_postprocessor
    from quantv2x_amd.ptq_state import export_ptq_state
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    contractive = model.endswith("-contractive")
    if model == "pyramid":
        qt = calibrated_pyramid_plugin(shape, n_agents=n_agents, n_points=n_points)
    else:
        qt = calibrated_plugin(shape, n_agents=n_agents, n_points=n_points, contractive=contractive)
    qt.model.hard_eval = True                                     # deterministic codebook pair (the wire format) in the mirror too
    sc = scene_np(n_agents, shape, n_points=n_points)
    with torch.no_grad():
        ref = qt(synth.scene_to_torch(sc))
        fp = None
        if model != "pyramid":                                    # the yardstick: the un-quantized mirror on the same scene
            from _common import build_plugin
            fpm = build_plugin(shape, contractive=contractive)
            fpm.hard_eval = True
            fp = fpm(synth.scene_to_torch(sc))
    lidar, vox = synth.SHAPES[shape][0], synth.SHAPES[shape][1]
    gw, gh, _ = synth.grid_size(lidar, vox)
    all_anchors = np.array(P.generate_anchor_boxes_3heads(lidar, gw, gh, MC_CFGS)[0])
    t = np.eye(4, dtype=np.float32)
    pp = build_postprocessor(mc_params(lidar, gw, gh), train=False)
    # (i) mirror -> CPU post-processor
    rb, rs, rl = P.post_process(ref["cls_preds"].numpy(), ref["reg_preds"].numpy(), None, interleave(all_anchors), t, pp.gt_range,
                                num_classes=3, max_extent=100.0, z_lim=(-100.0, 100.0), range_xy_only=True, return_labels=True)
    yard = {}
    if fp is not None:
        fb, fs, fl = P.post_process(fp["cls_preds"].numpy(), fp["reg_preds"].numpy(), None, interleave(all_anchors), t, pp.gt_range,
                                    num_classes=3, max_extent=100.0, z_lim=(-100.0, 100.0), range_xy_only=True, return_labels=True)
        if len(fs) and len(rs):
            yard = {"boxes_fp32_mirror": int(len(fs)),
                    "fp32_mirror_matched_in_w8a8_mirror_iou0.7": round(float(_match(fb, fl, rb, rl)[0].mean()), 4),
                    "w8a8_mirror_matched_in_fp32_mirror_iou0.7": round(float(_match(rb, rl, fb, fl)[0].mean()), 4)}
    # (ii) deployed path -> GPU post-processor
    eng = deploy(state=export_ptq_state(qt))
    out = eng(synth.scene_to_torch(sc, "cuda"))
    data = {"ego": {"transformation_matrix": torch.from_numpy(t), "all_anchors": torch.from_numpy(all_anchors), "num_anchors_per_location": [2, 2, 2]}}
    boxes, sl = pp.post_process(data, {"ego": out})
    gb = boxes.cpu().numpy() if boxes is not None else np.zeros((0, 8, 3), np.float32)
    gs = sl[:, 0].cpu().numpy() if boxes is not None else np.zeros(0, np.float32)
    gl = sl[:, 1].cpu().numpy().astype(np.int64) if boxes is not None else np.zeros(0, np.int64)
    assert len(rs) > 10 and len(gs) > 10
    m_g, arg_g, iou_g = _match(gb, gl, rb, rl)                    # deployed boxes found in the mirror's set
    m_r, _, _ = _match(rb, rl, gb, gl)                            # mirror boxes found in the deployed set
    ds = np.abs(gs[m_g] - rs[arg_g[m_g]])
    # raw head maps: how far the two paths drift before the detection logic
    dp = np.abs(out["preds_tensor"].cpu().numpy() - ref["preds_tensor"].numpy())
    top = min(50, len(gs), len(rs))
    m_top, _, _ = _match(gb[:top], gl[:top], rb, rl)
    report = {"model": model, "shape": shape, "agents": n_agents, "boxes_mirror": int(len(rs)), "boxes_deployed": int(len(gs)),
              "deployed_matched_in_mirror_iou0.7": round(float(m_g.mean()), 4), "mirror_matched_in_deployed_iou0.7": round(float(m_r.mean()), 4),
              f"top{top}_deployed_matched": round(float(m_top.mean()), 4),
              "score_abs_diff_matched_mean": round(float(ds.mean()), 5) if len(ds) else None,
              "score_abs_diff_matched_max": round(float(ds.max()), 5) if len(ds) else None,
              "mean_iou_of_matched": round(float(iou_g[m_g].mean()), 4) if m_g.any() else None,
              "preds_tensor_abs_diff_mean": round(float(dp.mean()), 5), "preds_tensor_abs_diff_p99": round(float(np.quantile(dp, 0.99)), 5),
              "head_lsb": round(max(float(qt.model.cls_head.act_quantizer.delta), float(qt.model.reg_head.act_quantizer.delta)), 5)}
    report.update(yard)
    print(json.dumps(report))
    outdir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(outdir):
        with open(os.path.join(outdir, f"box_agreement_{shape}.json" if model == "attfuse" else f"box_agreement_{model.replace('-', '_')}_{shape}.json"), "w") as f:
            json.dump(report, f, indent=1)
    # the bar: the two detection sets are the same objects
    bar = 0.95 if contractive else 0.6
    assert report["deployed_matched_in_mirror_iou0.7"] >= bar and report["mirror_matched_in_deployed_iou0.7"] >= bar, report
