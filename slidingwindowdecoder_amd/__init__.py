"""MI355X-native sliding-window QLDPC decoder.

Host-side problem construction (codes, circuit, windows) is pure numpy/scipy and importable
anywhere; the decoder classes need libswd_hip.so and a gfx950 GPU and raise otherwise.
"""
from .decoders import SlidingWindowDecoder, osd_window  # noqa: F401

__all__ = ["osd_window", "SlidingWindowDecoder"]
