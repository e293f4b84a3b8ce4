"""evae: host side of the MI355X-native Exemplar-VAE hot path (ctypes over libevae_hip.so)."""
from . import _lib  # noqa: F401
