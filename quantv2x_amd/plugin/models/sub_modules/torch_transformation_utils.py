"""``warp_affine_simple`` -- mirror of ``opencood/models/sub_modules/torch_transformation_utils.py:323-332``.

The reference accepts ``mode``/``padding_mode`` but does not forward them: the warp is always
bilinear with zero padding, ``align_corners`` as given (False by default)."""
import torch.nn.functional as F


def warp_affine_simple(src, M, dsize, mode='bilinear', padding_mode='zeros', align_corners=False):
    b, c = src.shape[0], src.shape[1]
    grid = F.affine_grid(M, [b, c, dsize[0], dsize[1]], align_corners=align_corners).to(src)
    return F.grid_sample(src, grid, align_corners=align_corners)
