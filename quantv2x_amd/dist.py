"""One agent per GPU: the V2X link as an RCCL all-gather of compressed code planes + poses.

The reference has no inference-time collective -- agents are rows of one batch on one GPU, the link is simulated
in-process (``heter_model_baseline.py:216`` stacks all agents, ``fusion_in_one.py:131-151`` regroups them) and the dataset
builds ``pairwise_t_matrix`` on the host from every agent's pose (``utils/transformation_utils.py:21-66``).  Here rank r
owns agent r and a step is

    pre   a1-a6 on the own agent (``frames`` frames per step)      -> payload: codes u8 [levels, frames, H*W] | poses f64 [frames, 4, 4]
    link  ONE fixed-size all-gather of the payload (RCCL over xGMI) -> gathered u8 [world, payload_bytes]
    post  pairwise matrix from the gathered poses, a7-a11 as the ego of this rank's own viewpoint (``ego = rank``), or only
          on rank 0 (``ego_only=True``: the parity configuration, identical to the single-process output)

``pre`` and ``post`` are each ONE HIP-graph replay once the input buffers are fixed; the collective sits between them on the
same stream.  105.6 KB of codes + 128 B of pose per agent-frame at V2X-Real shape (vs 36 MB of fp32 feature): the link is
latency-bound, not bandwidth-bound.

The collective is ``torch.distributed.all_gather_into_tensor`` by default (backend "nccl" IS RCCL on ROCm; "gloo" in the CPU
tests) or, with ``link="rccl"``, the C ABI's own ``qv2x_allgather_codes`` on a communicator created through
``qv2x_comm_init`` (the unique id is distributed with ``torch.distributed.broadcast_object_list``).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.distributed as dist

POSE_BYTES = 128            # one 4 x 4 float64 world pose


def payload_layout(levels: int, frames: int, hw: int):
    """(codes_bytes, pose_offset, payload_bytes) of one rank's payload; the pose block is 8-byte aligned."""
    codes = levels * frames * hw
    pose_off = (codes + 7) // 8 * 8
    return codes, pose_off, pose_off + frames * POSE_BYTES


def exchange_codes(codes: torch.Tensor, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One rank's contiguous payload -> [world, ...] (rank-major = agent-major) through torch.distributed."""
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
    """(agent_stride, level_stride) of bare all-gathered code planes [world, levels, hw] for ``qv2x_fuse_att_f32``."""
    return levels * hw, hw


class AgentShardedModel:
    """Drives a ``DeployedModel`` (or any object with the same stage interface: ``wire_shape``, ``encode_into``,
    ``pairwise_from_poses``, ``fuse_frames_and_heads``) with the agents of a scene sharded over ranks."""

    def __init__(self, engine, group=None, ego_only: bool = False, frames: int = 1, link: str = "torch", graphs: Optional[bool] = None,
                 max_cav: Optional[int] = None, graph_link: bool = False, emulate_world: Optional[int] = None,
                 emulate_poses: Optional[torch.Tensor] = None, emulate_rank: int = 0):
        if not getattr(engine, "has_codebook", True):
            raise NotImplementedError("AgentShardedModel exchanges the codebook's uint8 code planes: the codebook-less model "
                                      "has no compressed wire format (run it single-process through DeployedModel.forward)")
        if link not in ("torch", "rccl"):
            raise ValueError("link: 'torch' (torch.distributed's communicator) or 'rccl' (qv2x_allgather_codes)")
        self.engine, self.group, self.ego_only, self.frames, self.link = engine, group, ego_only, int(frames), link
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # REHEARSAL (bench.py --rehearse-world W [--rehearse-rank r]): one process plays rank ``emulate_rank`` of a world of W -- its own payload
        # fills every agent slot, the other agents' poses come from ``emulate_poses`` [W, 4, 4] -- so that the per-rank kernels of a W-GPU step
        # run with their true shapes, and with that rank's ego index (``ego = rank`` unless ``ego_only``), on one GPU.  Not a multi-GPU run:
        # no collective, and the "other agents" are copies of the own code planes.
        self.emulate = None
        if emulate_world is not None:
            if self.world != 1 or link != "torch" or emulate_poses is None or tuple(emulate_poses.shape) != (emulate_world, 4, 4):
                raise ValueError("emulate_world: one process, link='torch', emulate_poses [W, 4, 4]")
            if not 0 <= int(emulate_rank) < int(emulate_world):
                raise ValueError("emulate_rank: 0 <= r < emulate_world")
            self.world, self.emulate, self.rank = int(emulate_world), emulate_poses.to(torch.float64), int(emulate_rank)
        elif emulate_rank:
            raise ValueError("emulate_rank needs emulate_world")
        self.max_cav = max(self.world, max_cav or 0)
        self.levels, self.hw = engine.wire_shape()
        self.codes_bytes, self.pose_off, self.payload_bytes = payload_layout(self.levels, self.frames, self.hw)
        self.graphs = graphs
        # graph_link: the all-gather is captured INSIDE the step's HIP graph (pre + collective + post = ONE replay per step, no host
        # round trip between the stages).  Only for link="rccl": qv2x_allgather_codes is a plain ncclAllGather on the capturing stream.
        if graph_link and link != "rccl":
            raise ValueError("graph_link=True needs link='rccl' (the C ABI's own communicator)")
        self.graph_link = bool(graph_link)
        self._dev = None
        self._comm = None
        self._captured = None

    # ---- buffers ----------------------------------------------------------------------------------------------------------
    def _alloc(self, device):
        self._dev = device
        self.payload = torch.zeros(self.payload_bytes, dtype=torch.uint8, device=device)
        self.gathered = torch.zeros((self.world, self.payload_bytes), dtype=torch.uint8, device=device)
        self.pairwise = torch.zeros((self.frames, self.max_cav, self.max_cav, 4, 4), dtype=torch.float64, device=device)
        self.my_codes = self.payload[:self.codes_bytes].view(self.levels, self.frames, self.hw)
        self.my_poses = self.payload[self.pose_off:].view(torch.float64).view(self.frames, 4, 4)
        if self.graphs is None:
            self.graphs = device.type == "cuda"
        if self.link == "rccl":
            self._init_comm()

    def _init_comm(self):
        from . import lib as L
        lib = L.load()
        uid = (C.c_char * L.COMM_ID_BYTES)()
        if self.rank == 0:
            L.check(lib.qv2x_comm_unique_id(uid), "qv2x_comm_unique_id")
        box = [bytes(uid)]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=self.group)
        uid = (C.c_char * L.COMM_ID_BYTES).from_buffer_copy(box[0])
        comm = C.c_void_p()
        L.check(lib.qv2x_comm_init(uid, self.world, self.rank, C.byref(comm)), "qv2x_comm_init")
        self._comm, self._lib = comm, lib
        # The communicator's FIRST collective runs here, eagerly: RCCL sets up its channels and proxy connections lazily on a
        # communicator's first collective -- host-side allocations and synchronisations that must not happen under stream capture
        # (graph_link=True captures the step's all-gather).  Every rank is here together (the broadcast above), so the ranks stay in
        # phase; the gathered buffer it fills is overwritten by the first step.
        L.check(lib.qv2x_allgather_codes(comm, L.ptr(self.payload), L.ptr(self.gathered), self.payload_bytes, L.current_stream()),
                "qv2x_allgather_codes (communicator warm-up)")
        torch.cuda.current_stream().synchronize()

    def close(self):
        if self._comm is not None:
            self._lib.qv2x_comm_destroy(self._comm)
            self._comm = None

    # ---- the three phases ---------------------------------------------------------------------------------------------------
    def _pre(self, my_inputs: dict):
        self.engine.encode_into(my_inputs, self.frames, self.my_codes)

    def _set_pose(self, my_pose: torch.Tensor):
        """The step's pose goes straight into the payload, outside the graphs: a fresh pose tensor every step re-captures nothing."""
        self.my_poses.copy_(my_pose.to(torch.float64).expand(self.frames, 4, 4))

    def _exchange(self):
        if self.emulate is not None:
            self.gathered.copy_(self.payload.unsqueeze(0).expand(self.world, -1))
            self.gathered[:, self.pose_off:].view(torch.float64).view(self.world, self.frames, 4, 4).copy_(
                self.emulate.to(self.gathered.device)[:, None].expand(self.world, self.frames, 4, 4))
            return
        if self.link == "rccl":
            from . import lib as L
            L.check(self._lib.qv2x_allgather_codes(self._comm, L.ptr(self.payload), L.ptr(self.gathered), self.payload_bytes,
                                                   L.current_stream()), "qv2x_allgather_codes")
        else:
            exchange_codes(self.payload, self.group, self.gathered)

    def _post(self) -> dict:
        eng, ego = self.engine, (0 if self.ego_only else self.rank)
        if hasattr(eng, "pairwise_frames_from_poses"):                  # every frame's matrix in one launch
            eng.pairwise_frames_from_poses(self.gathered, self.world, self.payload_bytes, self.pose_off, self.frames, POSE_BYTES, self.max_cav,
                                           self.pairwise)
        else:
            for f in range(self.frames):
                eng.pairwise_from_poses(self.gathered, self.world, self.payload_bytes, self.pose_off + f * POSE_BYTES, self.max_cav, self.pairwise[f])
        return eng.fuse_frames_and_heads(self.gathered, self.payload_bytes, self.frames * self.hw, self.hw, self.pairwise,
                                         self.world, ego, self.my_codes, self.frames)

    # ---- one step -----------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, my_inputs: dict, my_pose: torch.Tensor) -> Optional[dict]:
        """``my_inputs``: the ``inputs_m1`` dict of THIS rank's agent, batch index = frame (0 .. frames-1); ``my_pose``: its 4 x 4
        world pose (``x_to_world(lidar_pose)``), one for the step or ``[frames, 4, 4]``.  Returns the model's output dict for
        ``frames`` scenes seen from this rank, or None on the non-ego ranks of ``ego_only``."""
        if self._dev is None:
            self._alloc(my_inputs["voxel_features"].device)
        skip_post = self.ego_only and self.rank != 0
        self._set_pose(my_pose)
        if not self.graphs:
            self._pre(my_inputs)
            self._exchange()
            return None if skip_post else self._post()
        key = (my_inputs["voxel_features"].data_ptr(), my_inputs["voxel_coords"].data_ptr(), my_inputs["voxel_num_points"].data_ptr(),
               tuple(my_inputs["voxel_features"].shape))
        if self._captured is None or self._captured[0] != key:
            # Capturing is RANK-LOCAL (no collective is issued by it, see _capture): a rank whose input buffers moved may re-capture
            # on its own without the ranks' all-gathers falling out of step.
            self._captured = (key,) + self._capture(my_inputs, skip_post)
        _, pre, post, out = self._captured
        pre.replay()
        if self.graph_link:                                          # one graph: a1-a6, the captured all-gather, a7-a11
            return out
        self._exchange()
        if post is not None:
            post.replay()
        return out

    def stage_graphs(self):
        """(pre, post) HIP graphs of the captured step (post is None on a non-ego rank of ``ego_only`` and with ``graph_link``), for stage
        timing; None before the first ``forward``."""
        if self._captured is None:
            return None
        return self._captured[1], self._captured[2]

    def _capture(self, my_inputs, skip_post):
        """Two HIP graphs around the collective.  The pillar count and the input addresses are baked in: refresh the input
        tensors in place between steps (pad unused pillar rows with agent index -1).

        No collective runs here -- neither in the warm-up nor between the two captures.  Every all-gather has the same size, so one
        issued by a rank that re-captures alone would pair with the other ranks' NEXT step and leave the ranks out of phase for good.
        The warm-up's post stage runs on this rank's own payload copied into every agent slot (valid codes and poses)."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                                   # warm-up outside capture (lazy one-off work: weight re-tiling, allocations)
                self._pre(my_inputs)
                self.gathered.copy_(self.payload.unsqueeze(0).expand(self.world, -1))
                if not skip_post:
                    self._post()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.graph_link:
            # ONE graph per step.  Capturing records the RCCL kernel without running it, so this is still rank-local; every rank's graph
            # holds exactly one all-gather, and the ranks replay in the same order.
            whole, out = torch.cuda.CUDAGraph(), None
            with torch.cuda.graph(whole):
                self._pre(my_inputs)
                self._exchange()
                if not skip_post:
                    out = self._post()
            return whole, None, out
        pre = torch.cuda.CUDAGraph()
        with torch.cuda.graph(pre):
            self._pre(my_inputs)
        post, out = None, None
        if not skip_post:
            post = torch.cuda.CUDAGraph()
            with torch.cuda.graph(post):
                out = self._post()
        return pre, post, out
