"""``VoxelPostprocessor`` of ``opencood/data_utils/post_processor/voxel_postprocessor.py`` for inference: same
constructor, ``generate_anchor_box`` (:30-83) and ``post_process(data_dict, output_dict)`` (:245-405) contract, the
work done by ``qv2x_postprocess_f32`` (``csrc/postprocess.hip``).  Training-side members (``generate_label``,
``collate_batch``, ``visualize``) are out of scope (datasets / training).

``post_process`` handles what intermediate / early fusion produces (``output_dict`` with the ego entry only) and late fusion (one
entry per CAV: ``qv2x_postprocess_late_f32``, up to 8 CAVs).  The head maps must live on the GPU (they come from ``DeployedModel``);
there is no CPU fallback.
"""
import ctypes as C
import math

import numpy as np
import torch

from ....lib import PostprocessDesc, check, current_stream, load, ptr


def load_point_pillar_anchor_args(hypes: dict) -> dict:
    """``yaml_utils.load_point_pillar_params`` (hypes_yaml/yaml_utils.py:106-146), the post-process part: fills
    ``vw, vh, vd, W, H, D`` of ``hypes['postprocess']['anchor_args']`` from the lidar range and the voxel size."""
    r = hypes["preprocess"]["cav_lidar_range"]
    vw, vh, vd = hypes["preprocess"]["args"]["voxel_size"]
    a = hypes["postprocess"]["anchor_args"]
    a.update(vw=vw, vh=vh, vd=vd, W=math.ceil((r[3] - r[0]) / vw), H=math.ceil((r[4] - r[1]) / vh), D=math.ceil((r[5] - r[2]) / vd))
    return hypes


def gpu_post_process(owner, cls, reg, dirp, anchors_dev, transformation_matrix, *, anchors_per_cell, num_classes, num_bins, dir_offset,
                     rng, range_xy_only, max_extent, z_lim, max_boxes, sync=True):
    """One ``qv2x_postprocess_f32`` call (or, when ``cls`` / ``reg`` / ``dirp`` / ``anchors_dev`` / ``transformation_matrix`` are lists with
    one entry per CAV, one ``qv2x_postprocess_late_f32`` call); ``owner`` keeps the workspace between frames.
    -> (corners [K, 8, 3], scores [K], labels i32 [K]) on the GPU, or (None, None, None)."""
    lib = load()
    many = isinstance(cls, (list, tuple))
    cls_l, reg_l = (list(cls), list(reg)) if many else ([cls], [reg])
    dir_l = (list(dirp) if many else [dirp]) if dirp is not None else [None] * len(cls_l)
    anc_l = list(anchors_dev) if many else [anchors_dev]
    t_l = list(transformation_matrix) if many else [transformation_matrix]
    ncav = len(cls_l)
    dev = cls_l[0].device
    d = PostprocessDesc()
    d.h, d.w, d.anchors_per_cell, d.num_bins = int(cls_l[0].shape[2]), int(cls_l[0].shape[3]), anchors_per_cell, num_bins
    d.score_threshold = float(owner.params["target_args"]["score_threshold"])
    d.nms_threshold = float(owner.params["nms_thresh"])
    d.dir_offset = dir_offset
    for i, v in enumerate(rng):
        d.range[i] = float(v)
    tf = (C.c_float * (16 * ncav))()
    for c, t in enumerate(t_l):
        t = np.asarray(t.detach().cpu() if torch.is_tensor(t) else t, dtype=np.float32).reshape(16)
        for i in range(16):
            tf[c * 16 + i] = float(t[i])
            if c == 0:
                d.transform[i] = float(t[i])
    d.max_boxes, d.num_classes, d.range_xy_only = max_boxes, num_classes, int(range_xy_only)
    d.max_extent, d.z_min, d.z_max = max_extent, z_lim[0], z_lim[1]
    need = lib.qv2x_postprocess_late_workspace_bytes(C.byref(d), ncav)
    if need < 0:
        check(-1, "qv2x_postprocess_late_workspace_bytes")
    if owner._ws is None or owner._ws.numel() < need or owner._ws.device != dev:
        owner._ws = torch.empty(need, dtype=torch.uint8, device=dev)
    corners = torch.empty((max_boxes, 8, 3), dtype=torch.float32, device=dev)
    scores = torch.empty((max_boxes,), dtype=torch.float32, device=dev)
    labels = torch.empty((max_boxes,), dtype=torch.int32, device=dev)
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    f32 = lambda x: x.to(torch.float32).contiguous()
    cls_l, reg_l = [f32(x) for x in cls_l], [f32(x) for x in reg_l]
    dir_l = [f32(x) if (x is not None and num_bins > 0) else None for x in dir_l]
    arr = lambda ts: (C.c_void_p * ncav)(*[(t.data_ptr() if t is not None else None) for t in ts])
    check(lib.qv2x_postprocess_late_f32(C.byref(d), ncav, arr(cls_l), arr(reg_l), arr(dir_l), arr(anc_l), tf, ptr(owner._ws), need,
                                        ptr(corners), ptr(scores), ptr(labels), ptr(count), current_stream()), "qv2x_postprocess_late_f32")
    if not sync:                                     # HIP-graph capture: fixed-size outputs + the box count left on the device
        return corners, scores, labels, count
    k = int(count.item())                            # the one host synchronisation of the frame (the reference goes to numpy here)
    if k == 0:
        return None, None, None
    return corners[:k], scores[:k], labels[:k]


class VoxelPostprocessor:
    def __init__(self, anchor_params, train):
        self.params = anchor_params
        self.train = train
        self.bbx_dict = {}
        self.anchor_num = self.params["anchor_args"]["num"]
        self._ws = None
        self._anchors_dev = None

    def generate_anchor_box(self):
        a = self.params["anchor_args"]
        r = [math.radians(e) for e in a["r"]]
        assert self.anchor_num == len(r)
        stride = a.get("feature_stride", 2)
        lr = a["cav_lidar_range"]
        x = np.linspace(lr[0] + a["vw"], lr[3] - a["vw"], a["W"] // stride)
        y = np.linspace(lr[1] + a["vh"], lr[4] - a["vh"], a["H"] // stride)
        cx, cy = np.meshgrid(x, y)
        cx = np.tile(cx[..., np.newaxis], self.anchor_num)
        cy = np.tile(cy[..., np.newaxis], self.anchor_num)
        cz = np.ones_like(cx) * -1.0
        w, l, h = np.ones_like(cx) * a["w"], np.ones_like(cx) * a["l"], np.ones_like(cx) * a["h"]
        r_ = np.ones_like(cx)
        for i in range(self.anchor_num):
            r_[..., i] = r[i]
        if self.params["order"] == "hwl":
            return np.stack([cx, cy, cz, h, w, l, r_], axis=-1)
        if self.params["order"] == "lhw":
            return np.stack([cx, cy, cz, l, h, w, r_], axis=-1)
        raise SystemExit("Unknown bbx order.")

    # ---- inference -------------------------------------------------------------------------------------------
    def post_process(self, data_dict, output_dict, max_boxes: int = 1000):
        """-> (pred_box3d_tensor [K, 8, 3], scores [K]) on the GPU, or (None, None) when nothing passes."""
        if self.params["order"] != "hwl":
            raise NotImplementedError("deployed post-process: box order 'hwl' (PointPillar)")
        cavs = [c for c in data_dict if c in output_dict]              # late fusion: one entry per CAV, the reference's loop order
        if not 1 <= len(cavs) <= 8:
            raise NotImplementedError("deployed post-process: 1..8 CAVs per call")
        cls_l, reg_l, dir_l, anc_l, t_l = [], [], [], [], []
        for cav_id in cavs:
            out, cav = output_dict[cav_id], data_dict[cav_id]
            cls = out["cls_preds"] if "cls_preds" in out else out["psm"]
            reg = out["reg_preds"] if "reg_preds" in out else out["rm"]
            dirp = out.get("dir_preds", out.get("dm"))
            if not cls.is_cuda:
                raise RuntimeError("VoxelPostprocessor.post_process runs on the GPU (libqv2x): the head maps must be CUDA tensors")
            if cls.shape[0] != 1 or cls.shape[1] != self.anchor_num:
                raise ValueError(f"cls_preds {tuple(cls.shape)}: batch 1 and {self.anchor_num} anchors per cell expected")
            h, w = int(cls.shape[2]), int(cls.shape[3])
            anchors = cav["anchor_box"]
            # device copies by anchor OBJECT (late fusion: every CAV carries its own `anchor_box`; one slot keyed on identity missed on
            # every CAV of every frame and re-uploaded H*W*A*7 floats each time); the entry keeps the object alive, so its id is not reused
            hit = self._anchors_dev.get(id(anchors)) if self._anchors_dev else None
            if hit is None or hit[0] is not anchors:
                a32 = torch.as_tensor(np.asarray(anchors.cpu() if torch.is_tensor(anchors) else anchors)).to(torch.float32)
                if tuple(a32.shape) != (h, w, self.anchor_num, 7):
                    raise ValueError(f"anchor_box {tuple(a32.shape)} does not match the head maps ({h}, {w}, {self.anchor_num}, 7)")
                if self._anchors_dev is None or len(self._anchors_dev) >= 16:
                    self._anchors_dev = {}
                hit = self._anchors_dev[id(anchors)] = (anchors, a32.reshape(-1, 7).contiguous().to(cls.device))
            cls_l.append(cls); reg_l.append(reg); dir_l.append(dirp); anc_l.append(hit[1]); t_l.append(cav["transformation_matrix"])
        has_dir = all(x is not None for x in dir_l)
        boxes, scores, _ = gpu_post_process(
            self, cls_l, reg_l, dir_l if has_dir else None, anc_l, t_l, anchors_per_cell=self.anchor_num,
            num_classes=1, num_bins=int(self.params["dir_args"]["num_bins"]) if has_dir else 0,
            dir_offset=float(self.params["dir_args"]["dir_offset"]) if has_dir else 0.0,
            rng=self.params["gt_range"], range_xy_only=False, max_extent=6.0, z_lim=(-3.0, 1.0), max_boxes=max_boxes)
        return boxes, scores
