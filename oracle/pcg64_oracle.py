"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the NumPy random-number machinery the reference's chains consume, for the
device generator of the 'pcg64' draw mode (mcmc_gpu_amd/csrc/pcg64_device.h, gsm_draw_pcg64).

The reference draws everything from numpy.random.Generator(PCG64) (gstatsMCMC/MCMC.py:483-492, :1056-1066); NumPy is a
third-party dependency (numpy 2.x here; the algorithms below have been stable since numpy 1.17), not part of the reference
tree, so its published algorithm is restated and pinned against NumPy ITSELF (tests/test_pcg64_oracle.py: bit-identical
streams, states included):

  PCG64                 128-bit LCG, multiplier 0x2360ed051fc65da44385df649fccf645, output XSL-RR 128/64
                        (numpy/random/src/pcg64/pcg64.h: pcg64_next64, pcg64_next32 with the has_uint32 / uinteger cache)
  next_double           (next_uint64 >> 11) * 2^-53
  Generator.random      next_double
  Generator.uniform     low + (high - low) * next_double              (distributions.c: random_uniform)
  Generator.integers    int64, [low, high): Lemire's bounded rejection on 32-bit words when the range fits 32 bits
                        (distributions.c: random_bounded_uint64_fill -> buffered_bounded_lemire_uint32)
  Generator.shuffle     Fisher-Yates from the back with random_interval: masked rejection on 32-bit words (distributions.c:
                        random_interval; _generator.pyx: shuffle)
  Generator.normal      loc + scale * standard_normal; standard_normal = 256-layer ziggurat (distributions.c:
                        random_standard_normal) with the tables of mcmc_gpu_amd/csrc/ziggurat_tables.h, log1p / exp of libm

`Pcg64` holds the state as Python integers and mirrors numpy's state dict (state, inc, has_uint32, uinteger).
"""
from __future__ import annotations

import math
import re
import struct
from pathlib import Path

MASK64 = (1 << 64) - 1
MASK128 = (1 << 128) - 1
PCG_MULT = 0x2360ED051FC65DA44385DF649FCCF645
ZIG_R = 3.6541528853610087963519472518
ZIG_INV_R = 0.27366123732975827203338247596


def load_tables():
    """(ki, wi, fi) from the generated header."""
    txt = (Path(__file__).resolve().parent.parent / "mcmc_gpu_amd" / "csrc" / "ziggurat_tables.h").read_text()
    words = [int(w, 16) for w in re.findall(r"0x([0-9a-f]{16})ull", txt)]
    assert len(words) == 768
    f = lambda w: struct.unpack("<d", struct.pack("<Q", w))[0]
    return words[:256], [f(w) for w in words[256:512]], [f(w) for w in words[512:]]


KI, WI, FI = load_tables()


class Pcg64:
    def __init__(self, state: int, inc: int, has_uint32: int = 0, uinteger: int = 0):
        self.state, self.inc, self.has_uint32, self.uinteger = int(state), int(inc), int(has_uint32), int(uinteger)
        self.n_raw = 0

    @classmethod
    def from_numpy(cls, gen):
        st = gen.bit_generator.state
        assert st["bit_generator"] == "PCG64"
        return cls(st["state"]["state"], st["state"]["inc"], st["has_uint32"], st["uinteger"])

    def numpy_state(self):
        return {"bit_generator": "PCG64", "state": {"state": self.state, "inc": self.inc}, "has_uint32": self.has_uint32,
                "uinteger": self.uinteger}

    # ---- bit generator -------------------------------------------------------------------------------------------------
    def next_uint64(self) -> int:
        self.state = (self.state * PCG_MULT + self.inc) & MASK128
        self.n_raw += 1
        hi, lo = self.state >> 64, self.state & MASK64
        x, rot = hi ^ lo, self.state >> 122
        return ((x >> rot) | (x << ((-rot) & 63))) & MASK64

    def next_uint32(self) -> int:
        if self.has_uint32:
            self.has_uint32 = 0
            return self.uinteger
        nxt = self.next_uint64()
        self.has_uint32 = 1
        self.uinteger = nxt >> 32
        return nxt & 0xFFFFFFFF

    def next_double(self) -> float:
        return (self.next_uint64() >> 11) * (1.0 / 9007199254740992.0)

    # ---- Generator methods -----------------------------------------------------------------------------------------------
    def random(self) -> float:
        return self.next_double()

    def uniform(self, low, high) -> float:
        return low + (high - low) * self.next_double()

    def integers(self, low: int, high: int) -> int:
        """Generator.integers(low, high, size=1)[0] for int64 and a range below 2^32 - 1."""
        rng = high - 1 - low
        if rng == 0:
            return low
        assert 0 < rng < 0xFFFFFFFF
        rng_excl = rng + 1
        m = self.next_uint32() * rng_excl
        leftover = m & 0xFFFFFFFF
        if leftover < rng_excl:
            threshold = (0xFFFFFFFF - rng) % rng_excl
            while leftover < threshold:
                m = self.next_uint32() * rng_excl
                leftover = m & 0xFFFFFFFF
        return low + (m >> 32)

    def interval(self, mx: int) -> int:
        """random_interval (distributions.c): uniform on [0, mx] by masked rejection on 32-bit words (mx < 2^32)."""
        if mx == 0:
            return 0
        mask = mx
        for sh in (1, 2, 4, 8, 16, 32):
            mask |= mask >> sh
        assert mx <= 0xFFFFFFFF
        while True:
            v = self.next_uint32() & mask
            if v <= mx:
                return v

    def shuffle(self, x):
        """Generator.shuffle(x) along axis 0 (_generator.pyx): for i = n - 1 .. 1: j = random_interval(i); swap rows i and j."""
        for i in reversed(range(1, len(x))):
            j = self.interval(i)
            if i == j:
                continue
            tmp = x[j].copy()
            x[j] = x[i]
            x[i] = tmp

    def standard_normal(self) -> float:
        while True:
            r = self.next_uint64()
            idx = r & 0xFF
            r >>= 8
            sign = r & 1
            rabs = (r >> 1) & 0x000FFFFFFFFFFFFF
            x = rabs * WI[idx]
            if sign:
                x = -x
            if rabs < KI[idx]:
                return x
            if idx == 0:
                while True:
                    xx = -ZIG_INV_R * math.log1p(-self.next_double())
                    yy = -math.log1p(-self.next_double())
                    if yy + yy > xx * xx:
                        return -(ZIG_R + xx) if ((rabs >> 8) & 1) else ZIG_R + xx
            elif (FI[idx - 1] - FI[idx]) * self.next_double() + FI[idx] < math.exp(-0.5 * x * x):
                return x

    def normal(self, loc=0.0, scale=1.0) -> float:
        return loc + scale * self.standard_normal()


def log1p_fdlibm(x: float) -> float:
    """log1p as glibc computes it for double (sysdeps/ieee754/dbl-64/s_log1p.c, the fdlibm algorithm) restated operation for
    operation: what the device evaluates in the ziggurat's tail, where the VALUE of log1p is returned (bit-identical to libm on
    the inputs the tail meets, x = -u with u in [0, 1): tests/test_pcg64_oracle.py)."""
    ln2_hi, ln2_lo = 6.93147180369123816490e-01, 1.90821492927058770002e-10
    Lp = (6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01,
          1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01)
    hx = struct.unpack("<q", struct.pack("<d", x))[0] >> 32
    ax = hx & 0x7FFFFFFF
    k, f, hu, c = 1, 0.0, 0, 0.0
    if hx < 0x3FDA827A:                                  # x < 0.41422
        if ax >= 0x3FF00000:                             # x <= -1
            return -math.inf if x == -1.0 else math.nan
        if ax < 0x3E200000:                              # |x| < 2**-29
            if ax < 0x3C900000:                          # |x| < 2**-54
                return x
            return x - x * x * 0.5
        if hx > 0 or hx <= struct.unpack("<i", struct.pack("<I", 0xBFD2BEC3))[0]:     # -0.2929 < x < 0.41422
            k, f, hu = 0, x, 1
    if hx >= 0x7FF00000:
        return x + x
    if k != 0:
        if hx < 0x43400000:
            u = 1.0 + x
            hu = struct.unpack("<q", struct.pack("<d", u))[0] >> 32
            k = (hu >> 20) - 1023
            c = (1.0 - (u - x)) if k > 0 else (x - (u - 1.0))
            c /= u
        else:
            u = x
            hu = struct.unpack("<q", struct.pack("<d", u))[0] >> 32
            k = (hu >> 20) - 1023
            c = 0.0
        hu &= 0x000FFFFF
        ub = struct.unpack("<q", struct.pack("<d", u))[0]
        if hu < 0x6A09E:
            u = struct.unpack("<d", struct.pack("<q", (ub & 0xFFFFFFFF) | ((hu | 0x3FF00000) << 32)))[0]
        else:
            k += 1
            u = struct.unpack("<d", struct.pack("<q", (ub & 0xFFFFFFFF) | ((hu | 0x3FE00000) << 32)))[0]
            hu = (0x00100000 - hu) >> 2
        f = u - 1.0
    hfsq = 0.5 * f * f
    if hu == 0:                                          # |f| < 2**-20
        if f == 0.0:
            if k == 0:
                return 0.0
            c += k * ln2_lo
            return k * ln2_hi + c
        R = hfsq * (1.0 - 0.66666666666666666 * f)
        if k == 0:
            return f - R
        return k * ln2_hi - ((R - (k * ln2_lo + c)) - f)
    s = f / (2.0 + f)
    z = s * s
    R1 = z * Lp[0]
    z2 = z * z
    R2 = Lp[1] + z * Lp[2]
    z4 = z2 * z2
    R3 = Lp[3] + z * Lp[4]
    z6 = z4 * z2
    R4 = Lp[5] + z * Lp[6]
    R = R1 + z2 * R2 + z4 * R3 + z6 * R4
    if k == 0:
        return f - (hfsq - s * (hfsq + R))
    return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f)
