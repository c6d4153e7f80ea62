"""GPU parity of bp4_osd against vectors recorded from the reference.  The device evaluates exp / log1p with the
algorithms of the C library the reference ran on (csrc/swd_libm.h, pinned by tests/test_libm_restatement.py), so
EVERY shot has to agree: error vectors, converge flags, iteration counts and OSD-0 solutions bit for bit; the
posterior LLRs are held to the north star's 1e-5 relative tolerance and the number of them that is bit-identical
is printed (all of them, when the goldens come from an FMA-capable glibc 2.35 host as committed)."""
import numpy as np
import pytest

from tests import fixtures as fx
from tests.test_oracle_bp4 import SHYPS_TAGS, TAGS, load_case, load_shyps

pytestmark = pytest.mark.gpu


def _check_all_shots(dec, out, c, label):
    same = (out == c["out"]).all(axis=(1, 2))
    conv = (dec.last_status & 0x100) != 0
    csame = conv == (c["converge"] != 0)
    isame = dec.last_iterations == c["its"]
    k = c["lpr"].shape[0]
    got = np.transpose(dec.last_llr[:k], (0, 2, 1))
    exact = (got == c["lpr"]) | (np.isnan(got) & np.isnan(c["lpr"]))
    print(f"{label}: {len(same)} shots, differing vectors {(~same).sum()}, converge flags {(~csame).sum()}, iteration counts "
          f"{(~isame).sum()}; posterior LLRs bit-identical {exact.sum()} of {exact.size}")
    assert same.all(), f"{label}: shots {np.flatnonzero(~same)[:10].tolist()} differ from the reference"
    assert csame.all() and isame.all()
    np.testing.assert_allclose(got, c["lpr"], rtol=1e-5, atol=1e-8)  # north star: within 1e-5 relative
    assert (dec.last_osd0 == c["osd0"]).all()
    return exact.mean()


@pytest.mark.parametrize("tag", TAGS)
def test_bp4_matches_reference(tag):
    from slidingwindowdecoder_amd import bp4_osd
    c = load_case(tag)
    dec = bp4_osd(c["code"].hx, c["code"].hz, channel_probs_x=c["pr"], channel_probs_y=c["pr"], channel_probs_z=c["pr"], **c["kw"])
    out = dec.decode_batch(c["sx"], c["sz"])
    assert _check_all_shots(dec, out, c, tag) == 1.0
    # decisions only (the posteriors, OSD-0 vectors and BP decisions stay on the device): the same vectors and status words
    st = dec.last_stats.copy()
    out2 = dec.decode_batch(c["sx"], c["sz"], details=False)
    assert np.array_equal(out2, out) and np.array_equal(dec.last_stats, st) and dec.last_llr is None and dec.last_osd0 is None


@pytest.mark.parametrize("tag", SHYPS_TAGS)
def test_bp4_shyps_matches_reference(tag):
    """BASELINE config 5: bp4_osd on the SHYPS r=3 stabiliser matrices (column weight 9), reference-recorded."""
    from slidingwindowdecoder_amd import bp4_osd, shyps
    c = load_shyps(tag)
    SX, SZ = shyps.shyps_stabilizers(3)
    dec = bp4_osd(SX, SZ, channel_probs_x=c["pr"], channel_probs_y=c["pr"], channel_probs_z=c["pr"], **c["kw"])
    assert (dec.rank_x, dec.rank_z) == (12, 12)
    out = dec.decode_batch(c["sx"], c["sz"])
    assert _check_all_shots(dec, out, c, "shyps " + tag) == 1.0
    cls = np.bincount(dec.last_status & 0xFF, minlength=6)
    assert cls[2] > 0  # the OSD ran on some shots


def test_bp4_single_decode_surface():
    from slidingwindowdecoder_amd import bp4_osd
    c = load_case("bb72_cs10")
    dec = bp4_osd(c["code"].hx, c["code"].hz, channel_probs_x=c["pr"], channel_probs_y=c["pr"], channel_probs_z=c["pr"], **c["kw"])
    for k in range(20):
        out = dec.decode(c["sx"][k], c["sz"][k])
        assert out.dtype == np.int64 and out.shape == (2, 72)
        assert (out == c["out"][k]).all() and dec.converge == c["converge"][k]
        # the returned correction reproduces both syndromes
        assert not ((c["code"].hx.astype(int) @ out[1] + c["sx"][k]) % 2).any()
        assert not ((c["code"].hz.astype(int) @ out[0] + c["sz"][k]) % 2).any()
    # bp_decoding_x/z and osdw_decoding_x/z (bp4_osd.pyx:606-620) on a shot where the OSD ran, vs the oracle
    from oracle import oracle as O
    ora = O.bp4_osd(c["code"].hx, c["code"].hz, channel_probs_x=c["pr"], channel_probs_y=c["pr"], channel_probs_z=c["pr"], **c["kw"])
    k = int(np.flatnonzero(c["converge"] == 0)[0])
    out, want = dec.decode(c["sx"][k], c["sz"][k]), ora.decode(c["sx"][k], c["sz"][k])
    assert (out == want).all() and dec.converge == 0
    assert (dec.bp_decoding_x == ora.bp_decoding_x).all() and (dec.bp_decoding_z == ora.bp_decoding_z).all()
    assert (dec.osdw_decoding_x == out[0]).all() and (dec.osdw_decoding_z == out[1]).all()
    with pytest.raises(ValueError):
        dec.decode(np.zeros(5), np.zeros(36))
    with pytest.raises(ValueError):
        bp4_osd(c["code"].hx, c["code"].hz[:, :-1], channel_probs_x=c["pr"], channel_probs_y=c["pr"], channel_probs_z=c["pr"])


def test_fuzz_random_codes_vs_oracle():
    """Random ragged Hx / Hz, priors and parameters (tests/fuzz_bp4.py, fixed seed)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_bp4.py"), "30", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("tag", ["bb72", "bb144"])
def test_bp4_camel_decode_matches_reference(tag):
    """camel_decode (bp4_osd.pyx:223-247) against the reference's recorded run.  A shot without a converged run
    returns what a new reference object returns (zeros); the recorded object returns its previous vectors there."""
    from slidingwindowdecoder_amd import bp4_osd
    from tests.test_oracle_bp4 import load_camel
    c = load_camel(tag)
    dec = bp4_osd(c["code"].hx, c["code"].hz, channel_probs_x=c["px"], channel_probs_y=c["py"], channel_probs_z=c["pz"], **c["kw"])
    out = dec.camel_decode_batch(c["sx"], c["sz"])
    conv = (dec.last_status & 0x100) != 0
    want = c["converge"] != 0
    assert (conv == want).all(), f"converge flags differ on shots {np.flatnonzero(conv != want)[:10].tolist()}"
    ok = conv & want
    same = (out == c["out"]).all(axis=(1, 2))
    assert same[ok].all(), f"{(~same[ok]).sum()} of {ok.sum()} converged shots differ"
    good = ok & same
    assert (dec.last_min_pm[good] == c["min_pm"][good]).all()
    assert np.array_equal(dec.last_iterations[good], c["its"][good])
    assert not out[~conv].any() and (dec.last_min_pm[~conv] == 10000.0).all()
    # single-call surface
    k = int(np.flatnonzero(good)[0])
    one = dec.camel_decode(c["sx"][k], c["sz"][k])
    assert one.dtype == np.int64 and (one == c["out"][k]).all() and dec.converge == 1 and dec.min_pm == dec.last_min_pm[0]
    assert (np.stack([dec.osd0_decoding_x, dec.osd0_decoding_z]) == one).all()


def test_bp4_concurrent_launches_of_one_handle_on_two_streams():
    """Round-4 advisor finding: the queue of unconverged decodes between the BP and the OSD kernel was one buffer per handle, so
    two launches of one handle on different streams raced for it (decodes dropped or solved against the other batch's pointers).
    The queue, the internal posterior buffer and the camel scratch now belong to a launch slot.  Many launches with a large
    share of OSD exits, alternating between two streams, must each equal the single-launch result."""
    import torch
    from slidingwindowdecoder_amd import bp4_osd
    c = load_case(TAGS[0])
    rng = np.random.default_rng(11)
    dec = bp4_osd(c["code"].hx, c["code"].hz, channel_probs_x=c["pr"], channel_probs_y=c["pr"], channel_probs_z=c["pr"], **c["kw"])
    # random (mostly unconverging) syndromes next to the recorded ones: the OSD queue is busy in every launch
    B = 2048
    sxs, szs, wants = [], [], []
    for k in range(2):
        sx = (rng.random((B, dec.mx)) < (0.02 + 0.03 * k)).astype(np.uint8)
        sz = (rng.random((B, dec.mz)) < (0.02 + 0.03 * k)).astype(np.uint8)
        sx[: len(c["sx"])] = c["sx"][: B]
        sz[: len(c["sz"])] = c["sz"][: B]
        out = dec.decode_batch(sx, sz, details=False)
        assert (np.bincount(dec.last_status & 0xFF, minlength=3)[2]) > B // 8
        sxs.append(sx); szs.append(sz); wants.append((out.copy(), dec.last_stats.copy()))
    dev = torch.device("cuda", 0)
    t_sx = [torch.from_numpy(x).to(dev) for x in sxs]
    t_sz = [torch.from_numpy(x).to(dev) for x in szs]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    torch.cuda.synchronize()
    res = []
    for i in range(12):  # more launches in flight than launch slots
        k = i & 1
        with torch.cuda.stream(streams[k]):
            res.append((k, dec.decode_batch_device(t_sx[k], t_sz[k], stream=streams[k])))
    torch.cuda.synchronize()
    for i, (k, (out, st)) in enumerate(res):
        assert np.array_equal(out.cpu().numpy(), wants[k][0]), f"launch {i} (stream {k}): vectors differ from the single launch"
        assert np.array_equal(st.cpu().numpy(), wants[k][1]), f"launch {i} (stream {k}): status words differ"


@pytest.mark.parametrize("tag", ["cs3", "e4"])
def test_bp4_unequal_ranks_matches_reference(tag):
    """rank(Hx) > rank(Hz) with a higher-order sweep (refused until round 6): the reference sizes both sweeps with kx = n - rank_x
    (bp4_osd.pyx:103-104, :284); vectors, converge flags and iteration counts of the run recorded from the reference.  The other
    direction -- the reference reads past its column array -- stays a refusal with a message."""
    from slidingwindowdecoder_amd import bp4_osd
    f = fx.load("bp4_unequal_ranks.npz")
    Hx, Hz = f[tag + "_hx"], f[tag + "_hz"]
    kw = fx.params(f, tag + "_params")
    pr = dict(channel_probs_x=f[tag + "_px"], channel_probs_y=f[tag + "_py"], channel_probs_z=f[tag + "_pz"])
    dec = bp4_osd(Hx, Hz, **pr, **kw)
    assert dec.rank_x > dec.rank_z
    sx, sz, want = fx.unpack(f[tag + "_sx"], Hx.shape[0]), fx.unpack(f[tag + "_sz"], Hz.shape[0]), fx.unpack(f[tag + "_out"], Hx.shape[1])
    out = dec.decode_batch(sx, sz)
    bad = np.flatnonzero((out != want).any(axis=(1, 2)))
    assert bad.size == 0, f"{bad.size} decodes differ: {bad[:8]}"
    assert np.array_equal((dec.last_status & 0x100) != 0, f[tag + "_converge"] != 0)
    assert np.array_equal(dec.last_iterations, f[tag + "_bp_iteration"])
    with pytest.raises((ValueError, RuntimeError), match="rank"):
        bp4_osd(Hz, Hx, **pr, **kw)  # rank(Hx) < rank(Hz)
