"""The C-ABI library loads on a CPU-only host and exports every symbol include/swd.h declares
(no compute calls here: decoding needs the GPU)."""
import os
import re

import pytest

from slidingwindowdecoder_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "swd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(swd_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_bound_and_exported():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libswd_hip.so not built (run __graft_entry__.build())")
    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 10
    bound = {s[0] for s in _lib.SYMBOLS}
    for nme in names:
        assert hasattr(L, nme), f"{nme} declared in include/swd.h but not exported"
        assert nme in bound, f"{nme} has no ctypes prototype in _lib.SYMBOLS"
    assert L.swd_abi_version() == 4


def test_no_cpu_fallback():
    """Without a GPU, constructing a decoder must fail loudly (never route to a CPU path)."""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libswd_hip.so not built")
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from slidingwindowdecoder_amd import osd_window
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        osd_window(np.eye(4, dtype=np.uint8), channel_probs=np.full(4, 0.1))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "slidingwindowdecoder_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert "swd_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn
