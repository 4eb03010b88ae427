"""One V2X-Real frame (2 agents) of the UN-quantized Pyramid model on the fp32 HIP engine.  ``python tools/bench_pyramid_fp32.py``"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantv2x_amd import synth  # noqa: E402
from quantv2x_amd.engine import deploy  # noqa: E402
from quantv2x_amd.plugin.tools import train_utils  # noqa: E402

model = train_utils.create_model(copy.deepcopy(synth.make_pyramid_hypes("v2xreal"))).eval()
synth.load_state_dict_numpy(model, synth.make_state_dict(model.state_dict(), seed=1))
eng = deploy(model)
dd = synth.scene_to_torch(synth.make_scene("v2xreal", n_agents=2, seed=0), "cuda")
for _ in range(3):
    eng(dd)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    eng(dd)
e1.record()
torch.cuda.synchronize()
print({"fp32_pyramid_ms_per_frame_2_agents": e0.elapsed_time(e1) / 10})
