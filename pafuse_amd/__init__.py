"""pafuse_amd - MI355X-native D3DP DDIM loop + per-part MixSTE denoisers behind the reference's module API."""
from .d3dp import D3DP
from .mixste2 import MixSTE2

__all__ = ["D3DP", "MixSTE2"]
