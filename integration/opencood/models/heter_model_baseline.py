import quantv2x_amd.plugin.models.heter_model_baseline as _m  # the MI355X-backed mirror of the reference module of this name (tools/make_opencood_shim.py)
globals().update({k: v for k, v in vars(_m).items() if not k.startswith('__')})
