"""uvltrack_amd/csrc/common.h::FastDiv -- the division by host-known values that replaced every integer division of the kernels' prologues (tile decodes, row
maps, LayerNorm rows, attention blocks): m = floor(2^32 / d) from the launcher (d = 1: 2^32 - 1), q = mulhi(x, m), one compare fixes q up.  Restated here in
exact integer arithmetic (the constants are READ from common.h so that the restatement cannot drift) and checked against x // d for every divisor shape the
launchers produce and for the edges of the 32-bit range: a wrong quotient would be a wrong tile or a wrong output row."""
import os
import random
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M32 = (1 << 32) - 1


def _source():
    return open(os.path.join(ROOT, "uvltrack_amd", "csrc", "common.h")).read()


def fastdiv_of(d):
    d = d if d else 1
    return d, (M32 if d == 1 else ((1 << 32) // d) & M32)


def fd_div(x, f):
    d, m = f
    q = (x * m) >> 32                      # __umulhi
    return q + 1 if ((x - q * d) & M32) >= d else q


def test_source_states_the_same_arithmetic():
    src = _source()
    assert "f.m = f.d == 1u ? 0xffffffffu : (uint32_t)((1ull << 32) / f.d);" in src
    assert re.search(r"const uint32_t q = __umulhi\(x, f\.m\);\s*return \(x - q \* f\.d >= f\.d\) \? q \+ 1 : q;", src)
    assert "uint32_t d = 1u << 30, m = 4;" in src and fastdiv_of(1 << 30) == (1 << 30, 4)      # the default matches GemmParams::rpb's "no row map"


def test_quotients_are_exact():
    rnd = random.Random(5)
    divisors = [1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 36, 40, 53, 55, 63, 96, 212, 220, 321, 361, 553, 681, 833, 873, 1024, 6664, 6984, 65535, 65536,
                (1 << 30) - 1, 1 << 30, (1 << 31) - 1, 1 << 31, M32 - 1, M32] + [rnd.randrange(1, 1 << 32) for _ in range(200)] + \
               [rnd.randrange(1, 5000) for _ in range(300)]
    for d in divisors:
        f = fastdiv_of(d)
        xs = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, M32, M32 - 1, (M32 // d) * d, max((M32 // d) * d - 1, 0)] + [rnd.randrange(0, 1 << 32) for _ in range(200)] + \
             [rnd.randrange(0, 1 << 20) for _ in range(200)]
        for x in xs:
            x &= M32
            assert fd_div(x, f) == x // d, (x, d)


def test_exhaustive_on_the_ranges_the_kernels_use():
    """tile / row / block indices below 2^16 against every divisor below 1200 (tiles per row of a grid, rows per sample, heads)"""
    for d in list(range(1, 1200)) + [6664, 6984, 17696]:
        f = fastdiv_of(d)
        for x in range(0, 1 << 16, 7):
            assert fd_div(x, f) == x // d
        for x in range(max(0, (1 << 16) - 3000), 1 << 16):
            assert fd_div(x, f) == x // d
