"""labelanything_amd - MI355X (gfx950) implementation of the LabelAnything inference hot path.

Host side (Python) mirrors the reference's module API (label_anything.models.LabelAnything);
all arithmetic runs in hand-written HIP kernels reached through the C ABI in include/la_hip.h.
"""
from .config import LamConfig, EncoderSpec, ENCODER_SPECS, register_encoder, config_from_kwargs  # noqa: F401

__version__ = "0.1.0"
