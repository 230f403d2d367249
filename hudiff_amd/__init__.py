"""hudiff_amd -- MI355X-native HuDiff sampling path (denoiser forward + T-step loop) behind a C ABI.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + C ABI), the ctypes binding, the
host-side mirror of the reference's model interface, the tokenizer / IMGT slot tables, the checkpoint
reader and the drop-in samplers.  See DESIGN.md.
"""
from .model import AntiTFNet, NanoAntiTFNet, model_selected, device_count, device_info  # noqa: F401

__all__ = ["AntiTFNet", "NanoAntiTFNet", "model_selected", "device_count", "device_info"]
