"""Single-class Pyramid model + residual multi-codebook compressor; mirror of ``opencood/models/heter_pyramid_collab_codebook.py``
(ctor ``:24-51``, codebook step ``:113-127``) -- ``heter_pyramid_collab_codebook_mc`` over ``HeterPyramidCollab``: ``channel = 64``,
``args['codebook'] = {seg_num, dict_size}`` (the OPV2V / DAIR yamls: 2 / 256), three residual levels.  ``hard_eval`` as in the mc mirror."""
from .heter_pyramid_collab import _named_modalities
from .heter_pyramid_collab_codebook_mc import HeterPyramidCollabCodebookMC


class HeterPyramidCollabCodebook(HeterPyramidCollabCodebookMC):
    def __init__(self, args):
        super().__init__(dict(args, num_class=1))

    def forward(self, data_dict):
        return super().forward(dict(data_dict, agent_modality_list=_named_modalities(self, data_dict['agent_modality_list'])))
