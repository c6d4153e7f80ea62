"""MI355X-native sliding-window QLDPC decoder.

Host-side problem construction (codes, circuit, windows) is pure numpy/scipy and importable
anywhere; the decoder classes need libswd_hip.so and a gfx950 GPU and raise otherwise.
"""
from .decoders import (SlidingWindowDecoder, bp4_osd, bp_history_decoder, bpgd_decoder,  # noqa: F401
                       bpgdg_decoder, osd_window)

__all__ = ["osd_window", "bpgdg_decoder", "bpgd_decoder", "bp_history_decoder", "bp4_osd", "SlidingWindowDecoder"]
