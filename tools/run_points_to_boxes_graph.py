"""Dev tool (GPU box): bench.py's points_to_boxes extra alone."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from quantv2x_amd.ptq_state import load_ptq_state
state = load_ptq_state(os.path.join(ROOT, "tools", "cache", "v2xreal_state.npz"))
print(json.dumps(bench.points_to_boxes_line(state, torch.device("cuda", 0)), indent=1))
