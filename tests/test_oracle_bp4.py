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
