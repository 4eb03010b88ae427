from .voxel_postprocessor import VoxelPostprocessor  # noqa: F401
from .voxel_postprocessor_3heads import VoxelPostprocessor3Heads  # noqa: F401

__all__ = {"VoxelPostprocessor": VoxelPostprocessor, "VoxelPostprocessor3Heads": VoxelPostprocessor3Heads}


def build_postprocessor(anchor_cfg, train):
    """``opencood/data_utils/post_processor/__init__.py:18-27``: lookup by ``core_method``."""
    name = anchor_cfg["core_method"]
    if name not in __all__:
        raise NotImplementedError(f"post-processor {name!r}: only the anchor-head post-processors are built "
                                  f"(Bev / CiaSSD / FPVRCNN / uncertainty: SURVEY.md §8(f))")
    return __all__[name](anchor_params=anchor_cfg, train=train)
