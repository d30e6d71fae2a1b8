"""Host-side binding of the MI355X-native Hybrid-ViT-Cascade hot path (see include/hvc_hip.h)."""
from . import _lib  # noqa: F401
