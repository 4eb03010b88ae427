"""HEAL Pyramid-fusion host model, SINGLE-class heads; mirror of ``opencood/models/heter_pyramid_collab.py`` (ctor ``:22-124``, forward
``:139-232``): the network of ``heter_pyramid_collab_mc`` with ``anchor_number`` / ``7 * anchor_number`` / ``num_bins * anchor_number`` head
channels and no ``num_class`` argument -- the ``core_method`` of the OPV2V / DAIR-V2X Pyramid yamls
(``hypes_yaml/opv2v/Codebook/Pyramid/pyramid_stage{2,3}_model.yaml``).  Same ``state_dict`` keys as the reference (the two classes differ
in head shapes only).  ``agent_modality_list`` may come as 1-based integer codes (``heter_pyramid_collab.py:141-152``)."""
import torch

from .heter_pyramid_collab_mc import HeterPyramidCollabMC


def _named_modalities(model, agent_modality_list):
    """the reference's remap of a tensor of 1-based modality codes onto ``modality_name_list`` (heter_pyramid_collab.py:143-150)"""
    if isinstance(agent_modality_list, torch.Tensor):
        return [model.modality_name_list[int(i) - 1] for i in agent_modality_list.tolist()]
    return agent_modality_list


class HeterPyramidCollab(HeterPyramidCollabMC):
    def __init__(self, args):
        super().__init__(dict(args, num_class=1))          # (c = 1: the mc head shapes a c c | 7 a c | bins a c are the single-class ones)

    def forward(self, data_dict):
        return super().forward(dict(data_dict, agent_modality_list=_named_modalities(self, data_dict['agent_modality_list'])))
