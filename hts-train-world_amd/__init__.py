"""hts-train-world_amd -- MI355X-native WORLD vocoder hot path (analysis + synthesis).

Host-side mirror of the reference's operator interface for this path
(externs/WORLD_v2/src/world/*.h): ``world.dio / stonemask / cheaptrick / d4c /
synthesis / harvest`` with the reference's argument meaning, plus the batched
device-resident ``WorldBatch``.  All compute happens in ``libworld_mi355.so``
(hand-written HIP for gfx950, C ABI in include/); Python and torch are plumbing
(device memory, streams, torch.distributed).  There is no CPU fallback: if the
library or a GPU is missing the calls raise.
"""
from . import world, synth_data, sharding, capi, recipe  # noqa: F401
from .world import WorldBatch, WorldParams, load_library  # noqa: F401
