"""Child process of tests/test_hip_dist_nccl.py (started fresh by torch.distributed.run, one per GPU, before anything touches a GPU):
rank r owns agent r of a tiny scene, the ranks exchange code planes + poses over RCCL, and rank 0 compares its fused output with the
single-process forward of the whole scene on its own GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    link = sys.argv[1] if len(sys.argv) > 1 else "torch"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from _common import calibrated_plugin, scene_np
    from oracle import geometry
    from quantv2x_amd import synth
    from quantv2x_amd.dist import AgentShardedModel
    from quantv2x_amd.engine import deploy
    from quantv2x_amd.ptq_state import export_ptq_state
    torch.set_num_threads(4)
    state = export_ptq_state(calibrated_plugin())
    eng = deploy(state=state, device=dev)
    sc = scene_np(world)
    co = sc["inputs_m1"]["voxel_coords"]
    mine = co[:, 0] == rank
    inp = {k: torch.from_numpy(v[mine].copy()).to(dev) for k, v in sc["inputs_m1"].items()}
    inp["voxel_coords"][:, 0] = 0
    poses = synth.agent_poses(world, "line")
    pose = torch.from_numpy(poses[rank]).to(dev)
    model = AgentShardedModel(eng, link=link)
    out = model.forward(inp, pose)
    out = {k: v.clone() for k, v in model.forward(inp, pose).items()}          # second step: the graph replay
    torch.cuda.synchronize()
    print(f"rank {rank} of {world}: RCCL world size {dist.get_world_size()}, link {link}", flush=True)
    if rank == 0:
        full = synth.scene_to_torch(sc, dev)
        full["pairwise_t_matrix"] = torch.from_numpy(geometry.pairwise_from_poses(poses, full["pairwise_t_matrix"].shape[1])[None]).to(dev)
        want = eng(full)
        torch.cuda.synchronize()
        for k in ("preds_tensor", "cls_preds", "reg_preds", "dir_preds"):
            assert torch.equal(out[k], want[k]), k
        assert torch.equal(out["cls_preds_single"][0], want["cls_preds_single"][0])
        print("OK sharded == single-process", flush=True)
    model.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
