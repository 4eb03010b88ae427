"""The Pyramid codebook model with the wire format made explicit; mirror of
``opencood/models/heter_pyramid_collab_codebook_mc_encdec.py`` (``encode_features :33-121``, ``decode_features :123-181``,
``forward_with_encdec :183-208``): ``encode_features`` = everything an agent runs before transmitting (-> code indices),
``decode_features`` = what the ego runs on the received codes."""
import torch

from ..utils.transformation_utils import normalize_pairwise_tfm
from .heter_pyramid_collab_codebook_mc import HeterPyramidCollabCodebookMC


class HeterPyramidCollabCodebookMCEncDec(HeterPyramidCollabCodebookMC):
    def encode_features(self, data_dict):
        agents = data_dict['agent_modality_list']
        affine = normalize_pairwise_tfm(data_dict['pairwise_t_matrix'], self.H, self.W, self.fake_voxel_size)
        feats = self.encode_agents(data_dict)
        n, c, h, w = feats.shape
        rows = feats.permute(0, 2, 3, 1).contiguous().view(-1, c)
        with torch.no_grad():
            codes = self.codebook.encode(rows)
        info = {'affine_matrix': affine, 'record_len': data_dict['record_len'], 'agent_modality_list': agents,
                'feature_shape': (n, c, h, w)}
        return codes, agents, info

    def decode_features(self, codes, other_info):
        n, c, h, w = other_info['feature_shape']
        feats = self.codebook.decode(codes).view(n, h, w, c).permute(0, 3, 1, 2).contiguous()
        return self.fuse_and_detect(feats, other_info['record_len'], other_info['affine_matrix'],
                                    other_info['agent_modality_list'], {'pyramid': 'collab'})

    def forward_with_encdec(self, data_dict):
        codes, _, info = self.encode_features(data_dict)
        return self.decode_features(codes, info)
