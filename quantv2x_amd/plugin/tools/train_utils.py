"""Model factory / device helpers of the inference boundary; mirror of
``opencood/tools/train_utils.py`` (``create_model :258-291``, ``to_device :394-403``,
``load_saved_model :171-219``)."""
import glob
import importlib
import os
import re

import torch

_MODEL_PACKAGE = "quantv2x_amd.plugin.models"


def create_model(hypes, package: str = _MODEL_PACKAGE):
    """``hypes['model']['core_method']`` names a module under ``package``; the class is found by
    case-insensitive name with underscores removed and built as ``cls(hypes['model']['args'])``.
    A miss prints the reference's message and exits with status 0 like the reference."""
    name = hypes['model']['core_method']
    module_name = package + "." + name
    wanted = name.replace('_', '')
    cls = None
    try:
        lib = importlib.import_module(module_name)
        for attr, obj in lib.__dict__.items():
            if attr.lower() == wanted.lower():
                cls = obj
    except ModuleNotFoundError:
        cls = None
    if cls is None:
        print('backbone not found in models folder. Please make sure you have a python file named %s '
              'and has a class called %s ignoring upper/lower case' % (module_name, wanted))
        exit(0)
    return cls(hypes['model']['args'])


def to_device(inputs, device):
    if isinstance(inputs, list):
        return [to_device(v, device) for v in inputs]
    if isinstance(inputs, dict):
        return {k: to_device(v, device) for k, v in inputs.items()}
    if isinstance(inputs, (int, float, str)) or not hasattr(inputs, 'to'):
        return inputs
    return inputs.to(device, non_blocking=True)


def load_saved_model(saved_path, model):
    """Load ``net_epoch_bestval_at*.pth`` if present, else the newest ``net_epoch*.pth``
    (``strict=False``, missing/unexpected keys reported).  Returns ``(epoch, model)``."""
    assert os.path.exists(saved_path), '{} not found'.format(saved_path)
    best = glob.glob(os.path.join(saved_path, 'net_epoch_bestval_at*.pth'))
    if best:
        assert len(best) == 1
        path = best[0]
        epoch = int(re.findall(r'bestval_at(\d+)', path)[0])
    else:
        epochs = [int(m[0]) for m in (re.findall(r'.*epoch(\d+)\.pth', f)
                                      for f in glob.glob(os.path.join(saved_path, '*epoch*.pth'))) if m]
        epoch = max(epochs) if epochs else 0
        path = os.path.join(saved_path, 'net_epoch%d.pth' % epoch)
        if epoch == 0:
            return 0, model
    state = torch.load(path, map_location='cpu')
    missing, unexpected = model.load_state_dict(state, strict=False)
    if missing:
        print(f"Missing keys from ckpt: {missing}")
    if unexpected:
        print(f"Unexpected keys from ckpt: {unexpected}")
    return epoch, model
