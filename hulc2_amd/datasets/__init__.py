"""Data side of the hot path: the play dataset resident in HBM, windows as index rows."""
from .device_store import DeviceEpisodeStore, fnv1_32, validation_window_size  # noqa: F401
