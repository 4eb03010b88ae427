"""MI355X-native implementation of QuantV2X's quantized per-agent encode + intermediate-fusion hot path.

    plugin/      host-side mirror of the reference's opencood.models / opencood.quant interfaces (PTQ surface)
    csrc/        HIP kernels for gfx950 + the C ABI (include/qv2x.h) -> libqv2x.so
    engine.py    deploy(qt_model) -> DeployedModel: the frozen W8A8 path on the GPU
    dist.py      one agent per GPU, RCCL all-gather of the code planes
    ptq_state.py frozen {delta, zero_point, uint8 weights} as plain arrays (save / load)
    synth.py     seeded synthetic scenes and weights (no dataset / checkpoint in this environment)
"""


def deploy(*args, **kwargs):
    from .engine import deploy as _deploy
    return _deploy(*args, **kwargs)
