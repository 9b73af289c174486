"""Julia's ``shuffle!(MersenneTwister(seed), v)`` reproduced bit for bit, for ``split`` (src/core.jl:15).

The reference seeds Julia's MersenneTwister -- the dSFMT-19937 generator of M. Saito and M. Matsumoto (the library
Julia links as libdSFMT 2.2) initialised by ``dsfmt_init_by_array`` on the 32-bit limbs of the seed -- and calls
``Random.shuffle!``: a Fisher-Yates pass from the back that draws ``j in 1..i`` by rejection from the low bits of the
52 mantissa bits of the next double in [1, 2) (``rand(r, ltm52(i, mask))`` with ``mask = nextpow(2, n) - 1``, halved
whenever ``mask >> 1 == i``).  This file restates exactly that (published algorithm + Julia 1.9 stdlib behaviour; the
reference pins Julia 1.9, Project.toml:24) and is pinned by the reference's own expected grouping for seed 1, k 5
(test/runtests.jl:31-32, kept in tests/golden/reference_kats.json) and by ``rand(MersenneTwister(1))`` =
0.23603334566204692.  Host-side arithmetic on Python integers; nothing here touches the GPU.
"""
from __future__ import annotations

from typing import List, Sequence

_M64 = (1 << 64) - 1
_M32 = (1 << 32) - 1
_N, _N64, _POS1, _SL1, _SR = 191, 382, 117, 19, 12
_MSK1, _MSK2 = 0x000FFAFFFFFFFB3F, 0x000FFDFFFC90FFFD
_FIX1, _FIX2 = 0x90014964B32F4329, 0x3B8D12AC548A7C7A
_PCV1, _PCV2 = 0x3D84E1AC0DC82880, 0x0000000000000001
_LOW_MASK, _HIGH_CONST = 0x000FFFFFFFFFFFFF, 0x3FF0000000000000


class MersenneTwister:
    """dSFMT-19937 seeded the way ``Random.MersenneTwister(seed::Integer)`` seeds it."""

    def __init__(self, seed: int):
        if seed < 0:
            raise ValueError("seed must be non-negative")
        key = []
        while True:                       # Random.make_seed: 32-bit limbs, least significant first
            key.append(seed & _M32)
            seed >>= 32
            if seed == 0:
                break
        size = (_N + 1) * 4
        lag = 11 if size >= 623 else 7 if size >= 68 else 5 if size >= 39 else 3
        mid = (size - lag) // 2
        p = [0x8B8B8B8B] * size
        f1 = lambda x: ((x ^ (x >> 27)) * 1664525) & _M32
        f2 = lambda x: ((x ^ (x >> 27)) * 1566083941) & _M32
        count = max(len(key) + 1, size)
        r = f1(p[0] ^ p[mid % size] ^ p[size - 1])
        p[mid % size] = (p[mid % size] + r) & _M32
        r = (r + len(key)) & _M32
        p[(mid + lag) % size] = (p[(mid + lag) % size] + r) & _M32
        p[0] = r
        count -= 1
        i = 1
        for j in range(count):
            r = f1(p[i] ^ p[(i + mid) % size] ^ p[(i + size - 1) % size])
            p[(i + mid) % size] = (p[(i + mid) % size] + r) & _M32
            r = (r + (key[j] if j < len(key) else 0) + i) & _M32
            p[(i + mid + lag) % size] = (p[(i + mid + lag) % size] + r) & _M32
            p[i] = r
            i = (i + 1) % size
        for _ in range(size):
            r = f2((p[i] + p[(i + mid) % size] + p[(i + size - 1) % size]) & _M32)
            p[(i + mid) % size] ^= r
            r = (r - i) & _M32
            p[(i + mid + lag) % size] ^= r
            p[i] = r
            i = (i + 1) % size
        s = [p[2 * t] | (p[2 * t + 1] << 32) for t in range(size // 2)]
        for t in range(2 * _N):
            s[t] = (s[t] & _LOW_MASK) | _HIGH_CONST
        inner = ((s[2 * _N] ^ _FIX1) & _PCV1) ^ ((s[2 * _N + 1] ^ _FIX2) & _PCV2)   # period certification
        sh = 32
        while sh:
            inner ^= inner >> sh
            sh >>= 1
        if not inner & 1:
            s[2 * _N + 1] ^= 1
        self._s = s
        self._buf: List[int] = []

    def _refill(self) -> None:
        s = self._s
        lung = [s[2 * _N], s[2 * _N + 1]]

        def rec(i: int, b: int) -> None:
            t0, t1 = s[2 * i], s[2 * i + 1]
            n0 = ((t0 << _SL1) & _M64) ^ (lung[1] >> 32) ^ ((lung[1] << 32) & _M64) ^ s[2 * b]
            n1 = ((t1 << _SL1) & _M64) ^ (lung[0] >> 32) ^ ((lung[0] << 32) & _M64) ^ s[2 * b + 1]
            lung[0], lung[1] = n0, n1
            s[2 * i] = (n0 >> _SR) ^ (n0 & _MSK1) ^ t0
            s[2 * i + 1] = (n1 >> _SR) ^ (n1 & _MSK2) ^ t1

        for i in range(_N - _POS1):
            rec(i, i + _POS1)
        for i in range(_N - _POS1, _N):
            rec(i, i + _POS1 - _N)
        s[2 * _N], s[2 * _N + 1] = lung
        self._buf = s[:_N64][::-1]

    def bits52(self) -> int:
        """The 52 mantissa bits of the next double in [1, 2) (Julia's ``rand(r, UInt52Raw())`` up to the exponent)."""
        if not self._buf:
            self._refill()
        return self._buf.pop() & _LOW_MASK

    def rand(self) -> float:
        """``rand(r)``: uniform in [0, 1)."""
        return self.bits52() / float(1 << 52)


def shuffle(items: Sequence, seed: int) -> list:
    """``shuffle!(MersenneTwister(seed), copy(items))`` (Random/src/misc.jl)."""
    a = list(items)
    n = len(a)
    if n <= 1:
        return a
    r = MersenneTwister(seed)
    mask = (1 << (n - 1).bit_length()) - 1        # nextpow(2, n) - 1
    for i in range(n, 1, -1):
        if (mask >> 1) == i:
            mask >>= 1
        while True:                                # rand(r, ltm52(i, mask)): masked rejection sampling of 0 .. i-1
            x = r.bits52() & mask
            if x < i:
                break
        a[i - 1], a[x] = a[x], a[i - 1]
    return a
