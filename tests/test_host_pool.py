"""The result-array pool of the host-buffer calls (decoders._HostPool): an array is only handed out again when the caller has
dropped every view of it, so recycling can never alias a result somebody still holds."""
import numpy as np

from slidingwindowdecoder_amd.decoders import _HostPool


def test_pool_recycles_only_dropped_arrays():
    p = _HostPool(per_shape=2)
    a = p.take((6, 7), np.uint8)
    a[:] = 3
    addr_a = a.ctypes.data
    b = p.take((6, 7), np.uint8)
    assert b.ctypes.data != addr_a                      # a is alive: a second buffer
    del a
    c = p.take((6, 7), np.uint8)
    assert c.ctypes.data == addr_a                      # dropped -> handed out again (warm pages)
    part = c[2:4, 1:3]                                  # a derived view keeps the buffer busy after the array itself is gone
    del c
    d = p.take((6, 7), np.uint8)
    assert d.ctypes.data != addr_a or not np.shares_memory(d, part)
    assert not np.shares_memory(d, part) and not np.shares_memory(d, b)
    e = p.take((6, 7), np.uint8)                         # beyond per_shape: plain fresh arrays, never an alias
    assert not any(np.shares_memory(e, x) for x in (b, d, part))
    f = p.take((6, 7), np.int32)                         # another dtype: its own buffers
    assert f.dtype == np.int32 and not np.shares_memory(f.view(np.uint8), b)


def test_pool_forgets_old_shapes():
    p = _HostPool(per_shape=1, shapes=2)
    for k in range(5):
        x = p.take((k + 1, 3), np.float64)
        assert x.shape == (k + 1, 3)
        del x
    assert len(p._bufs) <= 2


def test_pool_calibrates_its_idle_reference_count():
    """the pool measures what an idle base reads on this interpreter (3 on CPython 3.10) and falls back to fresh arrays when the
    probe disagrees with itself -- it never relies on a literal count"""
    p = _HostPool()
    assert p._idle is not None and p._probe(hold_view=True) == p._idle + 1
    p._idle = None                                       # an interpreter whose counts cannot be read: no recycling, no aliasing
    a = p.take((3, 3), np.uint8)
    addr = a.ctypes.data
    keep = a
    b = p.take((3, 3), np.uint8)
    assert not np.shares_memory(keep, b) and b.ctypes.data != addr and not p._bufs
