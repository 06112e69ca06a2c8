#!/usr/bin/env python3
"""Generate cymf_amd/csrc/mt_jump_poly.h: the MT19937 jump-ahead polynomial g(t) = t^J mod phi(t).

MT19937 is a linear recurrence over GF(2) on a 19937-bit state; phi is its characteristic
polynomial (degree 19937, found here with Berlekamp-Massey on one output bit), and
    state(n + J) = g(F) state(n) = XOR over the set bits k of g of state(n + k)
(Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer, "Efficient jump ahead for F2-linear random
number generators", 2008).  The device kernel (rng.hip: mt_jump_kernel) evaluates that XOR on the
recurrence's own word sequence, so only the 19937 coefficient bits are needed.  They depend on J
alone -- not on the seed.  Pure Python big-int arithmetic; runs in a few seconds.

    python tools/gen_mt_jump.py            # writes the header and self-checks against numpy's MT19937
"""
import os
import sys

import numpy as np

N, M = 624, 397
DEG = 19937
BLOCKS = 6720
J = N * BLOCKS            # words per chunk of the raw stream (4,193,280)
WIDE = 16                 # second polynomial: jump by WIDE chunks at once (the states of WIDE chunks follow from the WIDE before
                          # them in ONE launch of WIDE workgroups instead of a chain of WIDE one-workgroup launches)


def mt_words(state, count):
    """Untempered recurrence words x_0 .. x_{count-1}; x_0..x_623 = state."""
    x = list(int(v) for v in state)
    for k in range(count - N):
        y = (x[k] & 0x80000000) | (x[k + 1] & 0x7FFFFFFF)
        x.append(x[k + M] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0))
    return x


def berlekamp_massey(bits):
    """Minimal polynomial of a GF(2) sequence; polynomials as Python ints (bit i = coeff of t^i of
    the connection polynomial C, C_0 = 1)."""
    C, B, L, m = 1, 1, 0, 1
    srev = 0                                   # bit i = s_{n-i}
    for n, s in enumerate(bits):
        srev = (srev << 1) | s
        d = bin(C & srev).count("1") & 1       # sum_{i=0..L} C_i s_{n-i}
        if d == 0:
            m += 1
        elif 2 * L <= n:
            T = C
            C ^= B << m
            L, B, m = n + 1 - L, T, 1
        else:
            C ^= B << m
            m += 1
    return C, L


_SPREAD = [sum(((b >> i) & 1) << (2 * i) for i in range(8)) for b in range(256)]


def gf2_square(p):
    raw = p.to_bytes((p.bit_length() + 7) // 8 or 1, "little")
    out = bytearray(2 * len(raw))
    for i, b in enumerate(raw):
        v = _SPREAD[b]
        out[2 * i] = v & 0xFF
        out[2 * i + 1] = v >> 8
    return int.from_bytes(out, "little")


def gf2_mod(p, phi):
    d = phi.bit_length() - 1
    while p.bit_length() - 1 >= d:
        p ^= phi << (p.bit_length() - 1 - d)
    return p


def power_of_t(e, phi):
    """t^e mod phi by left-to-right square-and-multiply."""
    r = 1
    for bit in bin(e)[2:]:
        r = gf2_mod(gf2_square(r), phi)
        if bit == "1":
            r = gf2_mod(r << 1, phi)
    return r


def main():
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cymf_amd", "csrc", "mt_jump_poly.h")
    rs = np.random.RandomState(1234)
    s0 = rs.get_state()[1].astype(np.uint32)           # init_genrand(1234) state, pos = 624
    x = mt_words(s0, 2 * DEG + N + 8)
    bits = [x[k] >> 31 for k in range(1, 2 * DEG + 1)]  # one state bit along the orbit
    C, L = berlekamp_massey(bits)
    assert L == DEG, L
    # connection polynomial C(t) = sum C_i t^i  <->  characteristic polynomial phi(t) = t^L C(1/t)
    phi = sum(((C >> i) & 1) << (L - i) for i in range(L + 1))
    g = power_of_t(J, phi)
    assert g.bit_length() <= DEG
    gw = g
    for _ in range(WIDE.bit_length() - 1):              # t^(WIDE J) = g^WIDE, WIDE a power of two: repeated squaring
        gw = gf2_mod(gf2_square(gw), phi)
    assert 1 << (WIDE.bit_length() - 1) == WIDE and gw == power_of_t(WIDE * J, phi)

    # self-check on another seed: jump by the XOR formula == brute force
    rs = np.random.RandomState(987654321)
    st = rs.get_state()[1].astype(np.uint32)
    seq = mt_words(st, N + DEG)
    jumped = [0] * N
    for k in range(DEG):
        if (g >> k) & 1:
            for m in range(N):
                jumped[m] ^= seq[k + m]
    outs = rs.randint(0, 2**32, size=J, dtype=np.uint64).astype(np.uint32)   # consumes exactly J raw words
    want = rs.get_state()
    # after J = 624*BLOCKS outputs numpy's key array holds x_J .. x_{J+623} with pos = 624
    key = want[1].astype(np.uint32)
    assert int(want[2]) == 624, want[2]
    assert (jumped[0] >> 31) == (int(key[0]) >> 31) and jumped[1:] == [int(v) for v in key[1:]], "jump self-check failed"

    # the wide polynomial, checked the same way: WIDE * J outputs further on
    jumped_w = [0] * N
    for k in range(DEG):
        if (gw >> k) & 1:
            for m in range(N):
                jumped_w[m] ^= seq[k + m]
    for _ in range(WIDE - 1):
        rs.randint(0, 2**32, size=J, dtype=np.uint64)
    key = rs.get_state()[1].astype(np.uint32)
    assert (jumped_w[0] >> 31) == (int(key[0]) >> 31) and jumped_w[1:] == [int(v) for v in key[1:]], "wide jump self-check failed"

    words = [(g >> (32 * i)) & 0xFFFFFFFF for i in range(N)]
    words_w = [(gw >> (32 * i)) & 0xFFFFFFFF for i in range(N)]
    with open(out_path, "w") as f:
        f.write("// GENERATED by tools/gen_mt_jump.py -- do not edit.\n")
        f.write("// g(t) = t^J mod phi(t): MT19937 jump-ahead polynomial for J = 624 * %d raw words;\n" % BLOCKS)
        f.write("// bit k of the 19937-bit value (word k/32, bit k%32) is the coefficient of t^k.\n")
        f.write("#pragma once\n#include <cstdint>\nnamespace cymf {\n")
        f.write("constexpr int MT_JUMP_BLOCKS = %d;          // 624-word blocks per chunk\n" % BLOCKS)
        f.write("constexpr long long MT_JUMP_WORDS = %dLL;   // J\n" % J)
        f.write("static const uint32_t MT_JUMP_POLY[624] = {\n")
        for i in range(0, N, 8):
            f.write("    " + ", ".join("0x%08xu" % w for w in words[i:i + 8]) + ",\n")
        f.write("};\n")
        f.write("constexpr int MT_JUMP_WIDE = %d;              // chunks per wide jump\n" % WIDE)
        f.write("static const uint32_t MT_JUMP_POLY_WIDE[624] = {   // t^(MT_JUMP_WIDE * J) mod phi(t)\n")
        for i in range(0, N, 8):
            f.write("    " + ", ".join("0x%08xu" % w for w in words_w[i:i + 8]) + ",\n")
        f.write("};\n}  // namespace cymf\n")
    print("wrote", os.path.normpath(out_path), "J =", J, "popcount(g) =", bin(g).count("1"))


if __name__ == "__main__":
    sys.setrecursionlimit(10000)
    main()
