"""Calibrate the full-size synthetic model once on the CPU and cache the frozen PTQ state (dev tool)."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantv2x_amd import synth
from quantv2x_amd.plugin.tools import inference_quant, train_utils
from quantv2x_amd.ptq_state import export_ptq_state, save_ptq_state
shape = sys.argv[1] if len(sys.argv) > 1 else "v2xreal"
model = train_utils.create_model(copy.deepcopy(synth.make_hypes(shape))).eval()
synth.load_state_dict_numpy(model, synth.make_state_dict(model.state_dict(), seed=1))
calib = synth.scene_to_torch(synth.make_scene(shape, n_agents=1, seed=3, n_points=60000))
qt = inference_quant.calibrate_minmax(inference_quant.wrap(model), [calib])
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", f"{shape}_state.npz")
save_ptq_state(out, export_ptq_state(qt)); print("wrote", out)
