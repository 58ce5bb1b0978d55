"""The 8-byte gradient record of the table-gradient kernels (csrc/hashgrid_bwd.hip: pack_record /
unpack_record), restated in numpy: what survives the packing and how exactly.

13-bit slot | two values of (3-bit class c, 22-bit signed m): v ~ m 2^-(es0 + 4 c), es0 = 16 - E for a level
whose max |g| < 2^(E+1), c = min((E + 4 - exponent(v)) / 4, 7).  The kernels are checked against the oracle
on the GPU (tests/test_gpu_parity.py); this file pins the format's promises on the CPU.
"""
import numpy as np


def _bits(v):
    return int(np.float32(v).view(np.uint32))


def level_E(max_abs):
    return ((_bits(max_abs) >> 23) & 255) - 127


def rec_exponent(max_abs):
    return max(-126, min(16 - level_E(max_abs), 98))


def pack_value(v, E, es0):
    ev = ((_bits(v) >> 23) & 255) - 127
    c = min(max(E + 4 - ev, 0) >> 2, 7)
    scale = np.uint32((es0 + 4 * c + 127) << 23).view(np.float32)
    m = int(np.rint(np.float32(v) * scale))
    m = max(-2097151, min(m, 2097151))
    return (c << 22) | (m & 0x3FFFFF)


def pack_record(slot, v0, v1, E, es0):
    a, b = pack_value(v0, E, es0), pack_value(v1, E, es0)
    lo = ((slot & 0x1FFF) | ((a >> 22) << 13) | (a << 16)) & 0xFFFFFFFF
    hi = (((a >> 16) & 0x3F) | ((b >> 22) << 6) | ((b & 0x3FFFFF) << 9)) & 0xFFFFFFFF
    return lo, hi


def _sx22(x):
    x &= 0x3FFFFF
    return x - (1 << 22) if x >> 21 else x


def unpack_record(lo, hi):
    m0 = _sx22(((hi << 32 | lo) >> 16) & 0xFFFFFFFF)  # v_alignbit_b32(hi, lo, 16), low 22 bits
    return lo & 0x1FFF, m0, (lo >> 13) & 7, _sx22(hi >> 9), (hi >> 6) & 7


def decode(m, c, es0):
    return m * 2.0 ** -(es0 + 4 * c)


def test_round_trip_and_error_bounds():
    rng = np.random.default_rng(0)
    for max_abs in (3.7e-3, 0.998, 1.0, 5e-9, 812.0):
        E, es0 = level_E(max_abs), rec_exponent(max_abs)
        top = 2.0 ** (E + 1)
        worst_top, worst_rel = 0.0, 0.0
        for _ in range(4000):
            v0 = np.float32(max_abs * rng.uniform(-1, 1) * 10.0 ** rng.uniform(-9, 0))
            v1 = np.float32(max_abs * rng.uniform(-1, 1) * 10.0 ** rng.uniform(-3, 0))
            slot = int(rng.integers(0, 8192))
            s, m0, c0, m1, c1 = unpack_record(*pack_record(slot, v0, v1, E, es0))
            assert s == slot
            for v, m, c in ((v0, m0, c0), (v1, m1, c1)):
                r, v = decode(m, c, es0), float(v)
                assert c >= 1  # in-range values never use the overflow class
                if abs(v) >= 2.0 ** (E - 24):  # 18 .. 21 significant bits
                    worst_rel = max(worst_rel, abs(r - v) / abs(v))
                else:                          # fixed resolution 2^(E-44) below that
                    assert abs(r - v) <= 2.0 ** (E - 45)
                if abs(v) >= top / 16:
                    worst_top = max(worst_top, abs(r - v) / top)
        assert worst_rel <= 2.0 ** -18 and worst_top <= 2.0 ** -22, (max_abs, worst_rel, worst_top)


def test_extrapolated_weights_use_the_overflow_class():
    """A coordinate outside the grid has weights up to 2 per axis (the reference extrapolates): |w g| may
    reach 2^D max|g|, D <= 4 -- class 0 holds it instead of clamping."""
    max_abs = 0.9975
    E, es0 = level_E(max_abs), rec_exponent(max_abs)
    for w in (1.0, 1.728, 3.9, 15.9):
        v = np.float32(-0.8036797 * w)
        _, m, c, _, _ = unpack_record(*pack_record(5, v, 0.0, E, es0))
        assert abs(decode(m, c, es0) - float(v)) <= abs(float(v)) * 2.0 ** -18
        assert (c == 0) == (abs(float(v)) >= 2.0 ** (E + 1))


def test_zero_level_and_zero_values():
    E, es0 = level_E(0.0), rec_exponent(0.0)  # all-zero gradient: exponent field 0
    assert unpack_record(*pack_record(7, 0.0, -0.0, E, es0)) == (7, 0, 1, 0, 1)
    E, es0 = level_E(1.0), rec_exponent(1.0)
    _, m0, _, m1, _ = unpack_record(*pack_record(0, 1e-30, -1e-30, E, es0))
    assert m0 == 0 and m1 == 0  # below 2^(E-45): dropped
