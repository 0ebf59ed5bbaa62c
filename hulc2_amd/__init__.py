"""hulc2_amd — MI355X-native (gfx950) implementation of the HULC++ low-level policy training_step.

Host side mirrors the reference's Hydra/LightningModule interface (hulc2.models.*); compute is
hand-written HIP behind the C ABI in include/hulc2_amd.h.
"""
__version__ = "0.1.0"
