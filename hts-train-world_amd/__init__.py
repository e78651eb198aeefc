"""hts-train-world_amd -- MI355X-native WORLD vocoder hot path (analysis + synthesis).

Host-side mirror of the reference's operator interface for this path
(externs/WORLD_v2/src/world/*.h): ``world.dio / stonemask / cheaptrick / d4c /
synthesis / harvest`` with the reference's argument meaning, plus the batched
device-resident ``WorldBatch``.  All compute happens in ``libworld_mi355.so``
(hand-written HIP for gfx950, C ABI in include/); Python and torch are plumbing
(device memory, streams, torch.distributed).  There is no CPU fallback: if the
library or a GPU is missing the calls raise.
"""
from . import world, synth_data, sharding, capi  # noqa: F401
from .world import WorldBatch, WorldParams, load_library  # noqa: F401


def __getattr__(name):
    # `recipe` is also a command (python -m hts-train-world_amd.recipe): imported on first use, not with the package
    if name in ("recipe", "sweep", "pipeline"):
        import importlib
        return importlib.import_module(__name__ + "." + name)
    raise AttributeError(name)
