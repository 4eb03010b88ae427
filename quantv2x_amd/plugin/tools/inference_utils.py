"""Per-frame inference wrappers of the boundary; mirror of ``opencood/tools/inference_utils.py``
(``inference_early_fusion :123-194``, ``inference_intermediate_fusion :198-223``).

``model`` is anything with the reference's model contract -- the torch mirror, a ``QuantModel`` or a
``quantv2x_amd.DeployedModel``; ``dataset`` is the reference's dataset object (duck-typed: ``post_process`` and,
for two-item results, ``post_processor.generate_gt_bbx``).  The ONNX-wrapper branch of the reference is not mirrored
(NVIDIA-only deployment path).
"""
from collections import OrderedDict


def inference_early_fusion(batch_data, model, dataset):
    output_dict = OrderedDict()
    output_dict['ego'] = model(batch_data['ego'])
    assert isinstance(output_dict['ego'], dict), f"output_dict['ego'] must be a dict, got {type(output_dict['ego'])}"
    res = dataset.post_process(batch_data, output_dict)
    if isinstance(res, (list, tuple)) and len(res) == 3:
        pred_box_tensor, pred_score, gt_box_tensor = res
    else:
        pred_box_tensor, pred_score = res
        gt_box_tensor = dataset.post_processor.generate_gt_bbx(batch_data)
    out = {"pred_box_tensor": pred_box_tensor, "pred_score": pred_score, "gt_box_tensor": gt_box_tensor}
    if "depth_items" in output_dict['ego']:
        out["depth_items"] = output_dict['ego']['depth_items']
    return out


def inference_intermediate_fusion(batch_data, model, dataset):
    """Intermediate fusion happens inside the model; the per-frame call is the early-fusion one."""
    return inference_early_fusion(batch_data, model, dataset)
