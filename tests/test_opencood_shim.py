"""``integration/opencood``: the reference's package names re-exported from the mirror.  In a fresh
interpreter with ``integration/`` on ``sys.path`` (and the reference tree absent from it), the literal lookup of
opencood/tools/train_utils.py:272-291 -- ``importlib.import_module("opencood.models." + core_method)``, class by case-insensitive name
without underscores -- resolves every mirrored ``core_method``; ``opencood.quant`` / ``opencood.utils`` / ``opencood.tools`` names resolve
to the mirror's objects; a model built through the shim is the plugin's class."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CORE_METHODS = ["heter_model_baseline", "heter_model_baseline_mc", "heter_baseline_collab_codebook", "heter_baseline_collab_codebook_mc",
                "heter_pyramid_collab", "heter_pyramid_collab_mc", "heter_pyramid_collab_codebook", "heter_pyramid_collab_codebook_mc",
                "heter_pyramid_collab_codebook_mc_encdec"]

SCRIPT = r'''
import importlib, sys
assert not any("reference" in p for p in sys.path), sys.path
for core_method in %r:
    model_filename = "opencood.models." + core_method                      # train_utils.py:275-276
    model_lib = importlib.import_module(model_filename)
    target = core_method.replace('_', '')
    found = [cls for name, cls in model_lib.__dict__.items() if name.lower() == target.lower()]   # train_utils.py:280-282
    assert len(found) == 1, (core_method, found)
    mirror = importlib.import_module("quantv2x_amd.plugin.models." + core_method)
    assert found[0] is [c for n, c in mirror.__dict__.items() if n.lower() == target.lower()][0]
import opencood, os
assert os.path.dirname(opencood.__file__).endswith(os.path.join("integration", "opencood")), opencood.__file__
from opencood.quant import QuantModel, QuantModule, UniformAffineQuantizer, AdaRoundQuantizer, set_weight_quantize_params, set_act_quantize_params
from opencood.quant import layer_reconstruction, block_reconstruction, encoder_reconstruction
from opencood.quant.quant_layer import StraightThrough
from opencood.quant.fold_bn import search_fold_and_remove_bn
from opencood.utils.transformation_utils import normalize_pairwise_tfm, get_pairwise_transformation, x_to_world
from opencood.tools import train_utils, inference_utils
from opencood.tools.inference_utils import inference_intermediate_fusion
from opencood.models.sub_modules.codebook import UMGMQuantizer
from opencood.models.fuse_modules.fusion_in_one import AttFusion, MaxFusion
from opencood.data_utils.post_processor import build_postprocessor, VoxelPostprocessor
import quantv2x_amd.plugin.quant as q
assert QuantModel is q.QuantModel and UniformAffineQuantizer is q.UniformAffineQuantizer
# the reference's create_model, through the shim's own train_utils (its default package is the mirror; the literal package name works too)
from quantv2x_amd import synth
import copy
hy = synth.make_hypes("tiny")
m = train_utils.create_model(copy.deepcopy(hy), package="opencood.models")
assert type(m).__module__ == "quantv2x_amd.plugin.models." + hy["model"]["core_method"], type(m).__module__
print("shim ok")
'''


def test_reference_lookup_resolves_through_the_shim():
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "integration"), ROOT])
    r = subprocess.run([sys.executable, "-c", SCRIPT % (CORE_METHODS,)], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "shim ok" in r.stdout, r.stderr[-3000:]


def test_every_plugin_module_has_its_shim_file():
    src, dst = os.path.join(ROOT, "quantv2x_amd", "plugin"), os.path.join(ROOT, "integration", "opencood")
    for dirpath, _, files in os.walk(src):
        if "__pycache__" in dirpath:
            continue
        for f in files:
            if f.endswith(".py"):
                rel = os.path.relpath(os.path.join(dirpath, f), src)
                assert os.path.exists(os.path.join(dst, rel)), f"integration/opencood/{rel} missing: run tools/make_opencood_shim.py"
