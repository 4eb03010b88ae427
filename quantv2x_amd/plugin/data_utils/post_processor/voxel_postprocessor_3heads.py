"""``VoxelPostprocessor3Heads`` of ``opencood/data_utils/post_processor/voxel_postprocessor_3heads.py`` for inference
(the V2X-Real multi-class yaml): constructor, ``generate_anchor_box`` (:63-132) and
``post_process(data_dict, output_dict, projection=True)`` (:318-478) -> ``(pred_box3d_tensor [K, 8, 3],
score_labels [K, 2])``, the work done by ``qv2x_postprocess_f32`` with ``num_classes > 1``.
Training-side members are out of scope."""
import numpy as np
import torch

from .voxel_postprocessor import gpu_post_process

# opencood/data_utils/datasets/__init__.py:25 -- the module constant box_utils_mc.get_mask_for_boxes_within_range_torch reads
GT_RANGE = [-100, -40, -15, 100, 40, 15]


class VoxelPostprocessor3Heads:
    def __init__(self, anchor_params, train):
        self.params = anchor_params
        self.train = train
        self.bbx_dict = {}
        cfg = anchor_params["anchor_args"]["anchor_generator_config"]
        self.order = anchor_params["order"]
        self.anchor_generator_config = cfg
        self.anchor_sizes = [c["anchor_sizes"] for c in cfg]
        self.anchor_rotations = [c["anchor_rotations"] for c in cfg]
        self.anchor_heights = [c["anchor_bottom_heights"] for c in cfg]
        self.align_center = [c.get("align_center", False) for c in cfg]
        self.anchor_class_names = [c["class_name"] for c in cfg]
        self.matched_thresholds = {c["class_name"]: c["matched_threshold"] for c in cfg}
        self.unmatched_thresholds = {c["class_name"]: c["unmatched_threshold"] for c in cfg}
        assert len(self.anchor_sizes) == len(self.anchor_rotations) == len(self.anchor_heights)
        self.num_of_anchor_sets = len(self.anchor_sizes)
        self.grid_size = np.array([anchor_params["anchor_args"]["W"], anchor_params["anchor_args"]["H"]])
        self.cav_lidar_range = anchor_params["anchor_args"]["cav_lidar_range"]
        self.gt_range = list(GT_RANGE)
        self._ws = None
        self._anchors_dev = None

    def generate_anchor_box(self):
        r = self.cav_lidar_range
        grid_sizes = [self.grid_size[:2] // c["feature_map_stride"] for c in self.anchor_generator_config]
        all_anchors, num_anchors_per_location = [], []
        for gs, size, rot, height, centre in zip(grid_sizes, self.anchor_sizes, self.anchor_rotations, self.anchor_heights, self.align_center):
            num_anchors_per_location.append(len(rot) * len(size) * len(height))
            if centre:
                x_stride, y_stride = (r[3] - r[0]) / gs[0], (r[4] - r[1]) / gs[1]
                x_offset, y_offset = x_stride / 2, y_stride / 2
            else:
                x_stride, y_stride = (r[3] - r[0]) / (gs[0] - 1), (r[4] - r[1]) / (gs[1] - 1)
                x_offset, y_offset = 0, 0
            x_shifts = np.arange(r[0] + x_offset, r[3] + 1e-5, step=x_stride)
            y_shifts = np.arange(r[1] + y_offset, r[4] + 1e-5, step=y_stride)
            z_shifts = np.array(height)
            rot, size = np.array(rot), np.array(size)
            x_shifts, y_shifts, z_shifts = np.meshgrid(x_shifts, y_shifts, z_shifts)
            anchors = np.concatenate([x_shifts, y_shifts, z_shifts], axis=-1)
            a_size = np.tile(size.reshape(1, -1, 3), (*anchors.shape[0:2], 1))
            if self.order == "hwl":
                a_size = a_size[..., [2, 1, 0]]
            elif self.order == "lhw":
                a_size = a_size[..., [0, 2, 1]]
            else:
                raise SystemExit("Unknown bbx order.")
            anchors = np.concatenate((anchors, a_size), axis=-1)
            anchors = np.tile(anchors[:, :, None, :], (1, 1, len(rot), 1))
            a_rot = np.tile(rot.reshape(1, 1, -1, 1), (*anchors.shape[0:2], len(size), 1))
            all_anchors.append(np.concatenate([anchors, a_rot], axis=-1))
        return all_anchors, num_anchors_per_location

    def post_process(self, data_dict, output_dict, projection=True, max_boxes: int = 1000):
        if self.order != "hwl":
            raise NotImplementedError("deployed post-process: box order 'hwl' (PointPillar)")
        if not projection:
            raise NotImplementedError("deployed post-process returns the projected boxes (projection=True)")
        cavs = [c for c in data_dict if c in output_dict]              # late fusion: one entry per CAV, the reference's loop order (:345)
        if not 1 <= len(cavs) <= 8:
            raise NotImplementedError("deployed post-process: 1..8 CAVs per call")
        cls_l, reg_l, anc_l, t_l = [], [], [], []
        for cav_id in cavs:
            cav, out = data_dict[cav_id], output_dict[cav_id]
            cls, reg = out["cls_preds"], out["reg_preds"]
            if not cls.is_cuda:
                raise RuntimeError("VoxelPostprocessor3Heads.post_process runs on the GPU (libqv2x): the head maps must be CUDA tensors")
            all_anchors = cav["all_anchors"]                       # (num_class, H, W, anchor_num, 7)
            hit = self._anchors_dev.get(id(all_anchors)) if self._anchors_dev else None      # one device copy per anchor object (see VoxelPostprocessor)
            if hit is None or hit[0] is not all_anchors:
                a = torch.as_tensor(np.asarray(all_anchors.cpu() if torch.is_tensor(all_anchors) else all_anchors)).to(torch.float32)
                a = a.permute(1, 2, 0, 3, 4).contiguous()           # (H, W, num_class, anchor_num, 7), :354
                if self._anchors_dev is None or len(self._anchors_dev) >= 16:
                    self._anchors_dev = {}
                hit = self._anchors_dev[id(all_anchors)] = (all_anchors, a.reshape(-1, 7).to(cls.device), int(a.shape[2] * a.shape[3]), tuple(a.shape[:2]))
            _, anchors_dev, per_cell, hw = hit
            if cls.shape[0] != 1 or tuple(cls.shape[2:]) != hw or cls.shape[1] % per_cell or reg.shape[1] != per_cell * 7:
                raise ValueError(f"cls_preds {tuple(cls.shape)} / reg_preds {tuple(reg.shape)} do not match {per_cell} anchors per cell on {hw}")
            cls_l.append(cls); reg_l.append(reg); anc_l.append(anchors_dev); t_l.append(cav["transformation_matrix"])
        boxes, scores, labels = gpu_post_process(
            self, cls_l, reg_l, None, anc_l, t_l, anchors_per_cell=per_cell,
            num_classes=int(cls_l[0].shape[1] // per_cell), num_bins=0, dir_offset=0.0, rng=self.gt_range, range_xy_only=True,
            max_extent=100.0, z_lim=(-100.0, 100.0), max_boxes=max_boxes)
        if boxes is None:
            return None, None
        return boxes, torch.cat([scores.unsqueeze(1), labels.to(torch.float32).unsqueeze(1)], dim=1)
