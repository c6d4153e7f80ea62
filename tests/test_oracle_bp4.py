"""bp4_osd restatement in the oracle against vectors recorded from the reference extension
(tests/golden/make_golden.py bp4).  Equality is exact here because oracle and reference call the same
libm on the same machine; log/exp/log1p make BP4 results libm-dependent in general, which is why the
GPU tests compare the posterior LLRs with a tolerance."""
import numpy as np
import pytest

from oracle import oracle as O
from slidingwindowdecoder_amd.codes import bb_code
from tests import fixtures as fx

TAGS = ["bb72_cs10", "bb144_cs10", "bb72_e5", "bb144_osd0"]


def load_case(tag):
    f = fx.load("bp4_depolarizing.npz")
    N, p = int(f[tag + "_N"]), float(f[tag + "_p"])
    code, _, _ = bb_code(N)
    kw = fx.params(f, tag + "_params")
    pr = p / 3 * np.ones(N)
    mx, mz = code.hx.shape[0], code.hz.shape[0]
    return dict(code=code, kw=kw, pr=pr, sx=fx.unpack(f[tag + "_sx"], mx), sz=fx.unpack(f[tag + "_sz"], mz),
                out=fx.unpack(f[tag + "_out"], N), osd0=fx.unpack(f[tag + "_osd0"], N), converge=f[tag + "_converge"],
                its=f[tag + "_bp_iteration"], lpr=f[tag + "_lpr"])


@pytest.mark.parametrize("tag", TAGS)
def test_bp4_oracle_matches_reference(tag):
    c = load_case(tag)
    dec = O.bp4_osd(c["code"].hx, c["code"].hz, channel_probs_x=c["pr"], channel_probs_y=c["pr"], channel_probs_z=c["pr"], **c["kw"])
    for k in range(c["sx"].shape[0]):
        out = dec.decode(c["sx"][k], c["sz"][k])
        assert out.shape == (2, c["code"].N)
        assert (out == c["out"][k]).all(), f"decode {k}"
        assert dec.converge == c["converge"][k] and dec.bp_iteration == c["its"][k]
        assert (np.stack([dec.osd0_decoding_x, dec.osd0_decoding_z]) == c["osd0"][k]).all()
        if k < c["lpr"].shape[0]:
            np.testing.assert_allclose(dec.log_prob_ratios, c["lpr"][k], rtol=1e-12, atol=1e-12)


CAMEL_TAGS = ["bb72", "bb144"]


def load_camel(tag):
    f = fx.load("bp4_camel.npz")
    N = int(f[tag + "_N"])
    code, _, _ = bb_code(N)
    mx, mz = code.hx.shape[0], code.hz.shape[0]
    return dict(code=code, kw=fx.params(f, tag + "_params"), px=f[tag + "_px"], py=f[tag + "_py"], pz=f[tag + "_pz"],
                sx=fx.unpack(f[tag + "_sx"], mx), sz=fx.unpack(f[tag + "_sz"], mz), out=fx.unpack(f[tag + "_out"], N),
                converge=f[tag + "_converge"], its=f[tag + "_bp_iteration"], min_pm=f[tag + "_min_pm"])


@pytest.mark.parametrize("tag", CAMEL_TAGS)
def test_bp4_camel_decode_oracle_matches_reference(tag):
    """camel_decode (bp4_osd.pyx:223-247) in call order on one object: vectors, converge, min_pm bit for bit."""
    c = load_camel(tag)
    dec = O.bp4_osd(c["code"].hx, c["code"].hz, channel_probs_x=c["px"], channel_probs_y=c["py"], channel_probs_z=c["pz"], **c["kw"])
    for k in range(c["sx"].shape[0]):
        out = dec.camel_decode(c["sx"][k], c["sz"][k])
        assert (out == c["out"][k]).all(), f"decode {k}"
        assert dec.converge == c["converge"][k] and dec.bp_iteration == c["its"][k], f"decode {k}"
        assert dec.min_pm == c["min_pm"][k], f"decode {k}"


SHYPS_TAGS = ["osd0_p02", "osd0_p05", "cs10_p02", "cs10_p05"]


def load_shyps(tag):
    """bp4_osd on the SHYPS r=3 stabiliser matrices (tests/golden/make_golden.py bp4_shyps)."""
    f = fx.load("bp4_shyps.npz")
    SX, _ = fx.graph(f, "sx_")
    SZ, _ = fx.graph(f, "sz_")
    n = SX.shape[1]
    p = float(f[tag + "_p"])
    return dict(SX=SX, SZ=SZ, n=n, kw=fx.params(f, tag + "_params"), pr=p / 3 * np.ones(n),
                sx=fx.unpack(f[tag + "_sx"], SX.shape[0]), sz=fx.unpack(f[tag + "_sz"], SZ.shape[0]),
                out=fx.unpack(f[tag + "_out"], n), osd0=fx.unpack(f[tag + "_osd0"], n), converge=f[tag + "_converge"],
                its=f[tag + "_bp_iteration"], lpr=f[tag + "_lpr"])


def test_shyps_stabilizers_are_the_recorded_matrices():
    from slidingwindowdecoder_amd import shyps
    c = load_shyps("osd0_p02")
    SX, SZ = shyps.shyps_stabilizers(3)
    assert (c["SX"].toarray() == SX).all() and (c["SZ"].toarray() == SZ).all()
    assert SX.shape == (21, 49) and SX.sum(1).max() == 12 and SX.sum(0).max() == 9
    assert not (SX.astype(int) @ SZ.T % 2).any()


@pytest.mark.parametrize("tag", SHYPS_TAGS)
def test_bp4_shyps_oracle_matches_reference(tag):
    """BASELINE config 5's decoder on config 5's code: vectors, converge, iterations, OSD-0 solutions and posteriors
    of every recorded decode (posteriors exactly: oracle and reference call the same libm here)."""
    c = load_shyps(tag)
    dec = O.bp4_osd(c["SX"], c["SZ"], channel_probs_x=c["pr"], channel_probs_y=c["pr"], channel_probs_z=c["pr"], **c["kw"])
    for k in range(c["sx"].shape[0]):
        out = dec.decode(c["sx"][k], c["sz"][k])
        assert (out == c["out"][k]).all(), f"decode {k}"
        assert dec.converge == c["converge"][k] and dec.bp_iteration == c["its"][k]
        assert (np.stack([dec.osd0_decoding_x, dec.osd0_decoding_z]) == c["osd0"][k]).all()
        np.testing.assert_allclose(dec.log_prob_ratios, c["lpr"][k], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("tag", ["cs3", "e4"])
def test_bp4_unequal_ranks_oracle_matches_reference(tag):
    """rank(Hx) > rank(Hz) with a higher-order sweep (round-5 verdict, missing item 3): the reference sizes both sweeps with
    kx = n - rank_x (bp4_osd.pyx:103-104, :284); recorded from the reference extension by make_golden.py bp4_unequal"""
    f = fx.load("bp4_unequal_ranks.npz")
    Hx, Hz = f[tag + "_hx"], f[tag + "_hz"]
    dec = O.bp4_osd(Hx, Hz, channel_probs_x=f[tag + "_px"], channel_probs_y=f[tag + "_py"], channel_probs_z=f[tag + "_pz"], **fx.params(f, tag + "_params"))
    from slidingwindowdecoder_amd import gf2
    assert gf2.rank(Hx) > gf2.rank(Hz)
    sx, sz, out = fx.unpack(f[tag + "_sx"], Hx.shape[0]), fx.unpack(f[tag + "_sz"], Hz.shape[0]), fx.unpack(f[tag + "_out"], Hx.shape[1])
    for k in range(len(sx)):
        got = dec.decode(sx[k], sz[k])
        assert (got == out[k]).all(), f"decode {k}"
        assert dec.converge == f[tag + "_converge"][k] and dec.bp_iteration == f[tag + "_bp_iteration"][k]
    assert (f[tag + "_converge"] == 0).sum() > 100  # the sweeps really ran
