"""CPU check of the constants and flow graphs the HIP kernels hard-code (no GPU needed).

The kernels' Y transform is a SCALED 8-point DCT (csrc/common.hiph: dct8s; orthonormal X[k] = DS[k] * x[k]) whose
scales ride on the texture-feature sums.  A wrong constant there does not crash anything -- it biases the texture
mask by a few 1e-6 (this is how a mistyped cos(pi/16)/cos(3pi/16) was caught) -- so the numbers are re-derived here
from the source text and the flow graph is replayed in float32 against the oracle's float64 DCT."""
import math
import os
import re

import numpy as np

import offmark_oracle as orc
from conftest import PKG

F = np.float32
SRC = open(os.path.join(PKG, "csrc", "common.hiph")).read()


def const(name):
    m = re.search(r"\b" + name + r"\s*=\s*(-?[0-9.]+(?:e-?[0-9]+)?)f", SRC)
    assert m, name
    return float(m.group(1))


def fma(a, b, c):
    return (np.asarray(a, F).astype(np.float64) * np.float64(F(b)) + np.asarray(c, F).astype(np.float64)).astype(F)


def dct8s(x, T1, T2, T3, R13):
    """csrc/common.hiph dct8s, operation for operation, float32."""
    x = [x[..., i] for i in range(8)]
    a0, a1, a2, a3 = x[0] + x[7], x[1] + x[6], x[2] + x[5], x[3] + x[4]
    b0, b1, b2, b3 = x[0] - x[7], x[1] - x[6], x[2] - x[5], x[3] - x[4]
    c0, c1, c2, c3 = a0 + a3, a1 + a2, a1 - a2, a0 - a3
    y = [None] * 8
    y[0], y[4] = c0 + c1, c0 - c1
    y[2], y[6] = fma(c2, T2, c3), fma(c3, T2, -c2)
    p4, p7 = fma(b0, T3, b3), fma(b3, -T3, b0)
    p5, p6 = fma(b1, T1, b2), fma(b2, -T1, b1)
    u4, u6 = fma(p6, R13, p4), fma(p6, -R13, p4)
    u7, u5 = fma(p5, R13, p7), fma(p5, -R13, p7)
    y[1], y[7], y[3], y[5] = u7 + u4, u7 - u4, u5, u6
    return np.stack(y, -1).astype(F)


def test_constants_are_what_their_comments_say():
    c = lambda k: math.cos(k * math.pi / 16)            # noqa: E731
    want = dict(S0=math.sqrt(1 / 8), S1=0.5 * c(3) / math.sqrt(2), S2=0.5 * c(2), S3=0.5 * c(3),
                T1=math.tan(math.pi / 16), T2=math.tan(2 * math.pi / 16), T3=math.tan(3 * math.pi / 16), R13=c(1) / c(3),
                H1=0.5 * c(1), H2=0.5 * c(2), H3=0.5 * c(3), H4=0.5 * c(4), H5=0.5 * c(5), H6=0.5 * c(6), H7=0.5 * c(7),
                KY0=0.114, KY1=0.587, KY2=0.299, KU=0.492, KV=0.877, KDELTA=0.5, KI_B=2.032, KI_GU=-0.395, KI_GV=-0.581)
    for name, v in want.items():
        assert F(const(name)) == F(v), (name, const(name), v)      # the float32 the compiler sees is the nearest one


def test_scaled_dct_flow_graph_reproduces_the_orthonormal_dct():
    T1, T2, T3, R13 = (const(n) for n in ("T1", "T2", "T3", "R13"))
    DS = np.array([const(n) for n in ("S0", "S1", "S2", "S3", "S0", "S3", "S2", "S1")], F)
    rng = np.random.default_rng(1)
    blocks = rng.integers(0, 256, (4000, 8, 8)).astype(F)                  # Y-like data, worst case for cancellation
    blocks[:500] = rng.integers(100, 110, (500, 8, 8))                      # smooth blocks
    rows = dct8s(blocks, T1, T2, T3, R13)
    A = np.swapaxes(dct8s(np.swapaxes(rows, -1, -2), T1, T2, T3, R13), -1, -2)
    got = A.astype(np.float64) * DS[:, None].astype(np.float64) * DS[None, :].astype(np.float64)
    ref = orc.dct8x8(blocks).astype(np.float64)
    assert np.abs(got - ref).max() <= 1e-3                                   # the tests' Y-DC / coefficient tolerance
    assert np.abs(got - ref).max() <= 4e-7 * np.abs(ref).max()              # a few float32 ulp of the largest term
    # the texture-mask features from the scaled coefficients (kernel's grouping) against the oracle's
    ab = np.abs(A)
    S0, S1, S2, S3 = DS[0], DS[1], DS[2], DS[3]
    rs = fma(ab[..., 3, :] + ab[..., 5, :], S3, fma(ab[..., 2, :] + ab[..., 6, :], S2, fma(ab[..., 1, :] + ab[..., 7, :], S1,
             (ab[..., 0, :] + ab[..., 4, :]) * S0)))
    tot = fma(rs[..., 3] + rs[..., 5], S3, fma(rs[..., 2] + rs[..., 6], S2, fma(rs[..., 1] + rs[..., 7], S1, (rs[..., 0] + rs[..., 4]) * S0)))
    a00 = ab[..., 0, 0] * (S0 * S0)
    dcl = fma(ab[..., 2, 0], S2 * S0, fma(ab[..., 1, 1], S1 * S1, fma(ab[..., 1, 0] + ab[..., 0, 1], S1 * S0, fma(ab[..., 0, 2], S0 * S2, a00))))
    e = fma(ab[..., 3, 3], S3 * S3, fma(ab[..., 2, 2], S2 * S2, fma(ab[..., 2, 1] + ab[..., 1, 2], S2 * S1,
            fma(ab[..., 6, 0] + ab[..., 0, 6], S2 * S0, fma(ab[..., 5, 0] + ab[..., 0, 5], S3 * S0, fma(ab[..., 4, 0] + ab[..., 0, 4], S0 * S0,
                (ab[..., 3, 0] + ab[..., 0, 3]) * (S3 * S0)))))))
    dcl_o, eh_o, e_o = orc.texture_features(np.abs(orc.dct8x8(blocks)))
    # errors are a few float32 ulp of the block's total |coefficient| mass (eh = tot - dcl cancels on smooth blocks)
    rel = lambda x, y: np.abs(x.astype(np.float64) - y) / np.maximum(tot.astype(np.float64), 1.0)      # noqa: E731
    assert rel(tot - dcl, eh_o).max() <= 1e-6 and rel(dcl, dcl_o).max() <= 1e-6 and rel(e, e_o).max() <= 1e-6
    # no systematic bias (a mistyped constant shows up here first): mean signed relative error of eh
    assert abs(((tot - dcl).astype(np.float64) - eh_o).mean() / eh_o.mean()) <= 1e-7


def test_texture_thresholds_are_the_float32_images_of_the_float64_comparisons():
    """texture_mask compares float32 ratios with python floats, i.e. in float64 under the reference's numpy 1.23
    (dct_encoder.py:92-101).  The kernel compares in float32 against TA1/TB1/TA2/TB2 = the smallest float32 not below
    2.3 / 1.6 / 1.4 / 1.1: the same decision for every float32 input, nan and inf included."""
    for name, t in (("TA1", 2.3), ("TB1", 1.6), ("TA2", 1.4), ("TB2", 1.1)):
        m = re.search(r"\b" + name + r"\s*=\s*(0x[0-9a-fA-F.]+p[+-]?[0-9]+)f", SRC)
        assert m, name
        T = F(float.fromhex(m.group(1)))
        assert float(T) == float.fromhex(m.group(1))                       # the literal is a float32
        assert float(T) >= t > float(np.nextafter(T, F(-np.inf)))           # smallest float32 not below t
        x = T
        for _ in range(2000):                                               # 2000 float32 neighbours on each side
            x = np.nextafter(x, F(-np.inf))
        for _ in range(4000):
            assert (np.float64(x) >= t) == bool(x >= T)
            x = np.nextafter(x, F(np.inf))
        for special in (F(np.inf), F(-np.inf), F(np.nan), F(0), F(1e30)):
            with np.errstate(invalid="ignore"):
                assert (np.float64(special) >= t) == bool(special >= T)
