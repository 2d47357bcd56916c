"""CPU check of the integer GF(2^128) helpers the GHASH / XTS kernels are built
from (micro-aes_amd/csrc/uaes_gf.h compiles as host C++ too): wave-cooperative
multiply, multiplication tables, the strided-Horner level recursion and the
tweak shifts -- all against the oracle's bit-serial arithmetic."""
import ctypes as C
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gfc(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("gfc") / "libgfc.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", out,
                    os.path.join(ROOT, "tests", "csrc", "gf_check.cpp")], check=True)
    return C.CDLL(out)


def buf(b):
    return (C.c_uint8 * len(b)).from_buffer_copy(bytes(b))


def test_wave_cooperative_multiply(gfc, orc):
    rnd = random.Random(1)
    cases = [(bytes(16), rnd.randbytes(16)), (bytes([0x80] + [0] * 15), rnd.randbytes(16)),
             (bytes([0] * 15 + [1]), bytes([0] * 15 + [1])), (b"\xff" * 16, b"\xff" * 16)]
    cases += [(rnd.randbytes(16), rnd.randbytes(16)) for _ in range(200)]
    for x, y in cases:
        o = (C.c_uint8 * 16)()
        gfc.gfc_wave_mul(buf(x), buf(y), o)
        assert bytes(o) == orc.gf128_mul(x, y)


def test_tables_multiply_by_fixed_element(gfc, orc):
    rnd = random.Random(2)
    for _ in range(3):
        m = rnd.randbytes(16)
        t8, t4 = (C.c_uint8 * 65536)(), (C.c_uint8 * 8192)()
        gfc.gfc_table8(buf(m), t8)
        gfc.gfc_table4(buf(m), t4)
        for _ in range(50):
            a = rnd.randbytes(16)
            z8, z4 = (C.c_uint8 * 16)(), (C.c_uint8 * 16)()
            gfc.gfc_tabmul8(t8, buf(a), z8)
            gfc.gfc_tabmul4(t4, buf(a), z4)
            assert bytes(z8) == bytes(z4) == orc.gf128_mul(a, m)


def gf_pow(orc, h, e):
    r = bytes([0x80] + [0] * 15)
    for _ in range(e):
        r = orc.gf128_mul(r, h)
    return r


@pytest.mark.parametrize("n", [1, 5, 16, 17, 255, 256, 257, 1000])
def test_level_recursion_equals_ghash(gfc, orc, n):
    """strides 64 -> 16 -> 1 (small stand-ins for 2^17 -> 2^12 -> 256 -> 16 -> 1)"""
    rnd = random.Random(n)
    H = rnd.randbytes(16)
    data = rnd.randbytes(16 * n)
    # reference value: GHASH without the length block == orc.ghash minus last step
    acc = bytes(16)
    for k in range(n):
        acc = orc.gf128_mul(bytes(a ^ b for a, b in zip(acc, data[16 * k:16 * k + 16])), H)
    t64, t16, t1 = (C.c_uint8 * 65536)(), (C.c_uint8 * 8192)(), (C.c_uint8 * 8192)()
    gfc.gfc_table8(buf(gf_pow(orc, H, 64)), t64)
    gfc.gfc_table4(buf(gf_pow(orc, H, 16)), t16)
    gfc.gfc_table4(buf(H), t1)
    a64, a16, out = (C.c_uint8 * (64 * 16))(), (C.c_uint8 * (16 * 16))(), (C.c_uint8 * 16)()
    gfc.gfc_level(t64, 0, buf(data), C.c_uint64(n), C.c_uint64(64), a64)
    gfc.gfc_level(t16, 1, a64, C.c_uint64(64), C.c_uint64(16), a16)
    gfc.gfc_last(t1, a16, C.c_uint64(16), out)
    assert bytes(out) == acc


def test_xts_tweak_shifts(gfc, orc):
    """alpha^k by one shift == k doublings, cross-checked through the oracle's
    XTS: block j of a data unit is Enc(P ^ T*alpha^j) ^ T*alpha^j."""
    rnd = random.Random(3)

    def double(t):
        v = int.from_bytes(t, "little") << 1
        if v >> 128:
            v = (v & ((1 << 128) - 1)) ^ 0x87
        return v.to_bytes(16, "little")

    for _ in range(50):
        t = rnd.randbytes(16)
        cur = t
        for k in range(65):
            o = (C.c_uint8 * 16)()
            gfc.gfc_tw_pow(buf(t), k, o)
            assert bytes(o) == cur, k
            cur = double(cur)


def test_xts_chunk_tweak_by_sparse_powers(gfc):
    """tw_mul_a256(t, l) = t * alpha^(256 l) for every lane l = 0..63 (six conditional products by the sparse
    residues of alpha^(256 2^s)) against 256 l doublings"""
    rnd = random.Random(11)

    def double(v):
        v <<= 1
        return (v & ((1 << 128) - 1)) ^ 0x87 if v >> 128 else v

    for _ in range(6):
        t = rnd.randbytes(16)
        cur = int.from_bytes(t, "little")
        for l in range(64):
            o = (C.c_uint8 * 16)()
            gfc.gfc_tw_a256(buf(t), l, o)
            assert int.from_bytes(bytes(o), "little") == cur, l
            for _ in range(256):
                cur = double(cur)
