"""One agent per GPU: the V2X link as an RCCL all-gather of the compressed code planes.

The reference has no inference-time collective -- agents are rows of one batch on one GPU and the link is
simulated in-process (``heter_model_baseline.py:216`` stacks all agents).  Here rank r owns agent r:

    a1-a6  encode own agent            -> codes u8 [levels, H*W]        (no communication)
    link   all_gather_into_tensor      -> codes u8 [A, levels, H*W]     (one fixed-size collective: 105.6 KB/agent
                                                                          at V2X-Real shape vs 36 MB of fp32 feature)
    a7-a11 fuse + heads as the ego of rank r's own viewpoint (``ego = rank``), or only on rank 0
           (``ego_only=True``: the parity configuration, identical to the single-process output)

``torch.distributed`` with backend "nccl" is RCCL on ROCm; the CPU tests use "gloo" with a stand-in encoder.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


def exchange_codes(codes: torch.Tensor, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """codes u8 [levels, H*W] of this rank's agent -> [world, levels, H*W] (rank-major = agent-major)."""
    world = dist.get_world_size(group)
    codes = codes.contiguous()
    if out is None:
        out = torch.empty((world,) + tuple(codes.shape), dtype=codes.dtype, device=codes.device)
    if world == 1:
        out[0].copy_(codes)
    else:
        dist.all_gather_into_tensor(out, codes.unsqueeze(0), group=group)   # concat-along-dim-0 form (gloo and RCCL)
    return out


def gathered_strides(levels: int, hw: int):
    """(agent_stride, level_stride) of the all-gathered layout for ``qv2x_fuse_att_f32``."""
    return levels * hw, hw


class AgentShardedModel:
    """Drives a ``DeployedModel`` (or any object with the same three stage methods) with agents sharded over ranks."""

    def __init__(self, engine, group=None, ego_only: bool = False):
        if not getattr(engine, "has_codebook", True):
            raise NotImplementedError("AgentShardedModel exchanges the codebook's uint8 code planes: the codebook-less model "
                                      "has no compressed wire format (run it single-process through DeployedModel.forward)")
        self.engine = engine
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.ego_only = ego_only
        self._gathered = None

    @torch.no_grad()
    def forward(self, my_inputs: dict, pairwise_t_matrix: torch.Tensor) -> Optional[dict]:
        """``my_inputs``: the ``inputs_m1`` dict of THIS rank's agent (batch index 0);
        ``pairwise_t_matrix`` f64 [L, L, 4, 4] known to every rank (poses are exchanged with the codes in a real link)."""
        eng = self.engine
        codes = eng.encode_agents(my_inputs, 1)                      # [levels, 1, hw]
        levels, _, hw = codes.shape
        if self._gathered is None:
            self._gathered = torch.empty((self.world, levels, hw), dtype=codes.dtype, device=codes.device)
        gathered = exchange_codes(codes.view(levels, hw), self.group, self._gathered)
        ego = 0 if self.ego_only else self.rank
        if hasattr(eng, "fuse_heads_and_single") and not (self.ego_only and self.rank != 0):
            # fusion as the ego of this viewpoint + the *_single heads of this rank's own agent (one launch for both head passes)
            return eng.fuse_heads_and_single(gathered, *gathered_strides(levels, hw), pairwise_t_matrix, self.world, ego, codes, 1)
        single = eng.single_preds(codes, 1) if hasattr(eng, "single_preds") else {}     # this rank's own agent
        if self.ego_only and self.rank != 0:
            return None
        out = eng.fuse_and_heads(gathered, *gathered_strides(levels, hw), pairwise_t_matrix, self.world, ego)
        out.update(single)
        return out
