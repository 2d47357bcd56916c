"""Parity cases DERIVED from the table of arrangements.

Which kernels a call runs is decided by the planners of micro-aes_amd/csrc/uaes_plan.h; uaes_debug_plan() returns the
decision as data.  This file walks the sizes, finds every boundary b at which the decision (arrangement, GHASH
positions per thread) changes, and checks b - 16, b, b + 16 (and a ragged b - 3) against the oracle -- encryption,
decryption, and for the AEAD modes a forged tag -- so that a threshold that moves takes its tests with it.  Beyond
40 MiB, where the CPU oracle would need minutes, the same sizes are checked differentially: the arrangement the table
picks against the one that takes the call when that one is switched off (uaes_debug_plan_disable), head and tag
against the oracle.  The CPU half (table shape on a 256-CU device, names, the switch) is in tests/test_abi_and_host.py.
"""
import ctypes as C
import os
import random

import pytest

import micro_aes_amd as uaes

pytestmark = pytest.mark.gpu

MIB = 1 << 20
ORACLE_MAX = 40 * MIB


def regime(mode, n, b=0, direction=0, flags=0):
    name, _launches, _grid, steps = uaes.plan(mode, n, b, direction, flags)
    return name, steps


def sample_points(lo, hi, unit):
    """a log-dense grid of multiples of `unit` in [lo, hi]: 2^j - 1, 2^j, 2^j + 1 and 3 * 2^j units, and the first few"""
    ks = {0, 1, 2, 3, 5, 7}
    j = 1
    while (1 << j) * unit <= hi * 2:
        ks.update({(1 << j) - 1, 1 << j, (1 << j) + 1, 3 << (j - 1), 5 << (j - 2) if j > 1 else 1})
        j += 1
    return sorted(k * unit for k in ks if lo <= k * unit <= hi)


def boundaries(fn, lo, hi, unit=16):
    """[(b, regime below, regime from b on)]: every multiple of `unit` in (lo, hi] at which fn changes, found by bisection
    between the sample points"""
    pts = sample_points(lo, hi, unit)
    out = []
    for left, right in zip(pts, pts[1:]):
        rl, rr = fn(left), fn(right)
        while rl != rr:
            a, b = left, right                      # invariant: fn(a) == rl != fn(b)
            while b - a > unit:
                m = (a + b) // 2 // unit * unit
                if fn(m) == rl:
                    a = m
                else:
                    b = m
            out.append((b, rl, fn(b)))
            left, rl = b, fn(b)                     # there may be another change before `right`
    return out


def around(b, lo=0):
    return [n for n in (b - 16, b - 3, b, b + 16) if n >= lo]


def test_the_table_has_the_boundaries_the_design_names():
    """the planners on THIS device: every mode has its arrangements in the expected order, and the headline configs
    land where DESIGN.md says (C2 striped CTR, C3 bulk XTS, C4 striped GCM, tag-first decryption by the levels)"""
    assert uaes.plan("ctr", 1 << 30)[0] == "ctr.striped"
    assert uaes.plan("xts", 4096, 1 << 20)[0] == "xts.bulk"
    assert uaes.plan("xts", 4096, 1024)[0] == "xts.small" and uaes.plan("xts", 512, 1 << 16)[0] == "xts.packed"
    assert uaes.plan("gcm", 1 << 30)[0] == "gcm.striped" and uaes.plan("gcm", 1 << 30)[1] == 3
    assert uaes.plan("gcm", 1 << 30, 0, 1)[0] == "gcm.levels"
    assert uaes.plan("ecb", 4096)[0] == "ecb.single"
    names = [r[2][0] for r in boundaries(lambda n: regime("gcm", n), 0, 200 * MIB)]
    assert [n for i, n in enumerate(names) if i == 0 or names[i - 1] != n] == ["gcm.chunks", "gcm.twophase", "gcm.striped"], names
    names = [r[2][0] for r in boundaries(lambda n: regime("gcm", n, 0, 1), 0, 600 * MIB)]
    assert [n for i, n in enumerate(names) if i == 0 or names[i - 1] != n] == ["gcm.chunks", "gcm.levels"], names
    assert [r[2][0] for r in boundaries(lambda n: regime("ctr", n), 0, 64 * MIB)] == ["ctr.quad", "ctr.striped"]
    assert [r[2][0] for r in boundaries(lambda n: regime("ecb", n), 0, 64 * MIB)] == ["ecb.tiled"]
    assert [r[2][0] for r in boundaries(lambda n: regime("ocb", n), 0, 64 * MIB)] == ["ocb.runs"]


def test_ecb_and_ctr_at_every_boundary(orc):
    rnd = random.Random(601)
    for bits in (128, 256):
        key, iv = rnd.randbytes(bits // 8), rnd.randbytes(12)
        for b, below, above in boundaries(lambda n: regime("ecb", n), 0, 64 * MIB):
            for n in (b - 16, b, b + 16):
                data = orc.splitmix(n, n)
                ct = uaes.AES_ECB_encrypt(key, data)
                assert ct == orc.ecb_encrypt(key, data), ("ecb", n, below, above)
                assert uaes.AES_ECB_decrypt(key, ct) == (0, data), ("ecb", n)
        for b, below, above in boundaries(lambda n: regime("ctr", n), 0, 64 * MIB):
            for n in around(b):
                data = orc.splitmix(n + 1, n)
                assert uaes.AES_CTR_encrypt(key, iv, data) == orc.ctr_encrypt(key, iv, data), ("ctr", n, below, above)


def test_xts_at_every_boundary(orc):
    rnd = random.Random(602)
    keys = rnd.randbytes(64)
    # one data unit of n bytes (the reference's call shape, micro_aes.c:1066-1093), explicit tweak and sector-0 tweak
    for flags, tweak in ((2, rnd.randbytes(16)), (0, None)):
        for b, below, above in boundaries(lambda n: regime("xts", max(n, 16), 1, 0, flags), 16, 24 * MIB):
            for n in around(b, 16):
                data = orc.splitmix(n + 2, n)
                rc, ct = uaes.AES_XTS_encrypt(keys, tweak, data)
                assert (rc, ct) == orc.xts(keys, tweak, data, True, prefill=0), ("xts unit", n, below, above)
                assert uaes.AES_XTS_decrypt(keys, tweak, ct) == (0, data), ("xts unit", n)
    # many units: the boundary in the NUMBER of units, for sector sizes on either side of a chunk
    import torch
    L = uaes.engine()
    ks = rnd.randbytes(32)
    for sector in (512, 4096, 4096 + 16, 8192):
        found = boundaries(lambda k: regime("xts", sector, max(k, 1)), 1, (160 * MIB) // sector, unit=1)
        for b, below, above in found:
            for ns in (b - 1, b, b + 1):
                if ns < 1 or ns * sector > 192 * MIB:
                    continue
                n = ns * sector
                src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0")
                dst = torch.empty(n, dtype=torch.uint8, device="cuda:0")
                uaes.xts_sectors_dev(ks, 7, sector, ns, src, dst)
                torch.cuda.synchronize()
                # head, middle and tail units against the oracle (each unit stands alone: tweak = its number)
                for u in sorted({0, 1, ns // 2, ns - 1}):
                    if u >= ns:
                        continue
                    unit = bytes(src[u * sector:(u + 1) * sector].cpu().numpy())
                    want = orc.xts_sectors(ks, 7 + u, sector, unit, True)[1]
                    assert bytes(dst[u * sector:(u + 1) * sector].cpu().numpy()) == want, ("xts units", sector, ns, u, below, above)
                back = torch.empty_like(src)
                uaes.xts_sectors_dev(ks, 7, sector, ns, dst, back, encrypt=False)
                torch.cuda.synchronize()
                assert torch.equal(back, src), ("xts units", sector, ns)
                del src, dst, back


def _aead_case(orc, enc, dec, oracle_enc, n, aad, what, cache=None):
    data = orc.splitmix(n + 5, n)
    if cache is not None and (n, aad) in cache:
        want = cache[(n, aad)]                      # (the oracle is the slow part: one encryption per text, not per direction)
    else:
        want = oracle_enc(aad, data)
        if cache is not None:
            cache[(n, aad)] = want
    got = enc(aad, data)
    assert got == want, what
    assert dec(aad, got) == (0, data), what
    bad = bytearray(got)
    bad[(n // 2) if n else len(bad) - 1] ^= 0x10
    rc = dec(aad, bytes(bad))[0]
    assert rc == 0x1A, what
    if n <= (1 << 20):
        bad = bytearray(got)
        bad[-1] ^= 0x01
        assert dec(aad, bytes(bad))[0] == 0x1A, what


def test_gcm_at_every_boundary_vs_the_oracle(orc):
    """both default directions (encrypt; decrypt = tag first, N7) and the opt-in one-pass decryption, at every boundary
    of each direction's table up to ORACLE_MAX, with and without associated data"""
    L = uaes.engine()
    rnd = random.Random(603)
    key, nonce = rnd.randbytes(16), rnd.randbytes(12)
    aads = {0: b"", 37: rnd.randbytes(37)}
    seen, wants = set(), {}
    try:
        for direction in (0, 1, 2):
            L.uaes_set_gcm_one_pass_decrypt(1 if direction == 2 else 0)
            for alen in (0, 37):
                found = boundaries(lambda n: regime("gcm", n, alen, direction), 0, ORACLE_MAX)
                assert len(found) >= 4, found
                for b, below, above in found:
                    for n in around(b):
                        if (direction, alen, n) in seen:
                            continue
                        seen.add((direction, alen, n))
                        _aead_case(orc, lambda a, d: uaes.AES_GCM_encrypt(key, nonce, a, d),
                                   lambda a, c: uaes.AES_GCM_decrypt(key, nonce, a, c),
                                   lambda a, d: orc.gcm_encrypt(key, nonce, a, d), n, aads[alen],
                                   ("gcm", direction, alen, n, below, above), wants)
    finally:
        L.uaes_set_gcm_one_pass_decrypt(0)


def test_gcm_beyond_the_oracle_against_the_next_arrangement(orc):
    """the boundaries past ORACLE_MAX (64 positions per thread, two phases -> striped at 128 MiB, the tag-first
    decryption's chunk workgroups up to 512 MiB): the planned arrangement against the one that takes the call when it
    is switched off -- two independent sets of kernels must agree on every byte and on the tag --, the first 64 KiB and
    its keystream against the oracle, a forged tag refused with the output untouched"""
    import torch
    L = uaes.engine()
    rnd = random.Random(604)
    key, nonce = rnd.randbytes(16), rnd.randbytes(12)
    status = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
    try:
        for direction, hi in ((0, 200 * MIB), (1, 600 * MIB)):
            found = [f for f in boundaries(lambda n: regime("gcm", n, 0, direction), 0, hi) if f[0] > ORACLE_MAX]
            assert found, (direction, found)
            for b, below, above in found:
                for n in (b - 16, b, b + 16):
                    src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0")
                    planned = uaes.plan("gcm", n, 0, direction)[0]
                    ct = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0")
                    uaes.gcm_encrypt_dev(key, nonce, None, src, n, ct)
                    torch.cuda.synchronize()
                    head = 1 << 16
                    assert bytes(ct[:head].cpu().numpy()) == orc.gcm_encrypt(key, nonce, b"", bytes(src[:head].cpu().numpy()))[:head]
                    if direction == 0:
                        L.uaes_debug_plan_disable(1 << uaes.arrangement_id(planned))
                        other = uaes.plan("gcm", n, 0, 0)[0]
                        assert other != planned
                        ct2 = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0")
                        uaes.gcm_encrypt_dev(key, nonce, None, src, n, ct2)
                        torch.cuda.synchronize()
                        L.uaes_debug_plan_disable(0)
                        assert torch.equal(ct, ct2), (n, planned, other)
                        del ct2
                    back = torch.full((n,), 0xCC, dtype=torch.uint8, device="cuda:0")
                    uaes.gcm_decrypt_dev(key, nonce, None, ct, n, back, status)
                    torch.cuda.synchronize()
                    assert int(status.item()) == 0 and torch.equal(back, src), (direction, n, planned)
                    ct[n // 3] ^= 2
                    back.fill_(0xCC)
                    uaes.gcm_decrypt_dev(key, nonce, None, ct, n, back, status)
                    torch.cuda.synchronize()
                    assert int(status.item()) == 0x1A and int((back != 0xCC).sum()) == 0, (direction, n, planned)
                    del src, ct, back
    finally:
        L.uaes_debug_plan_disable(0)


def test_every_gcm_arrangement_takes_every_size_it_can(orc):
    """the arrangements behind the first choice, reached by switching the ones in front of them off (the table's
    fall-through order): levels for a 100 KiB text, striped for a 20 MiB one, chunks without the small kernel ... --
    each against the oracle, encryption and both decryption orders"""
    L = uaes.engine()
    rnd = random.Random(605)
    key, nonce, aad = rnd.randbytes(32), rnd.randbytes(12), rnd.randbytes(21)
    ids = {n: uaes.arrangement_id(n) for n in ("gcm.small", "gcm.chunks", "gcm.twophase", "gcm.striped")}
    cases = [((1 << ids["gcm.small"]), 3000, "gcm.chunks"),
             ((1 << ids["gcm.small"]) | (1 << ids["gcm.chunks"]), 3000, "gcm.levels"),
             ((1 << ids["gcm.chunks"]), 100 * 1024 + 5, "gcm.levels"),
             ((1 << ids["gcm.chunks"]), 9 * MIB + 16, "gcm.striped"),
             ((1 << ids["gcm.chunks"]) | (1 << ids["gcm.striped"]), 9 * MIB + 16, "gcm.levels"),
             ((1 << ids["gcm.twophase"]), 20 * MIB + 48, "gcm.striped"),
             ((1 << ids["gcm.twophase"]) | (1 << ids["gcm.striped"]), 20 * MIB + 48, "gcm.levels")]
    try:
        for mask, n, want_arr in cases:
            L.uaes_debug_plan_disable(mask)
            assert uaes.plan("gcm", n, len(aad))[0] == want_arr, (hex(mask), n)
            for one_pass in (0, 1):
                L.uaes_set_gcm_one_pass_decrypt(one_pass)
                _aead_case(orc, lambda a, d: uaes.AES_GCM_encrypt(key, nonce, a, d),
                           lambda a, c: uaes.AES_GCM_decrypt(key, nonce, a, c),
                           lambda a, d: orc.gcm_encrypt(key, nonce, a, d), n, aad, ("gcm", hex(mask), n, want_arr, one_pass))
    finally:
        L.uaes_debug_plan_disable(0)
        L.uaes_set_gcm_one_pass_decrypt(0)


def test_ocb_and_gcmsiv_at_every_boundary(orc):
    L = uaes.engine()
    rnd = random.Random(606)
    key, nonce = rnd.randbytes(16), rnd.randbytes(12)
    for alen in (0, 70000):                                   # long associated data moves OCB's boundary
        for b, below, above in boundaries(lambda n: regime("ocb", n, alen), 0, 24 * MIB):
            for n in around(b):
                aad = rnd.randbytes(alen)
                _aead_case(orc, lambda a, d: uaes.AES_OCB_encrypt(key, nonce, a, d),
                           lambda a, c: uaes.AES_OCB_decrypt(key, nonce, a, c),
                           lambda a, d: orc.ocb_encrypt(key, nonce, a, d), n, aad, ("ocb", alen, n, below, above))
    sivs = boundaries(lambda n: regime("siv", n, 9), 0, ORACLE_MAX)
    assert sivs and sivs[0][2][0] == "siv.chunks", sivs
    for b, below, above in sivs:
        for n in around(b):
            data, aad = orc.splitmix(n + 9, n), rnd.randbytes(9)
            want = orc.gcmsiv_encrypt(key, nonce, aad, data)
            assert uaes.GCM_SIV_encrypt(key, nonce, aad, data) == want, ("siv", n, below, above)
            assert uaes.GCM_SIV_decrypt(key, nonce, aad, want) == (0, data), ("siv", n)
            bad = bytearray(want)
            bad[-3] ^= 8
            assert uaes.GCM_SIV_decrypt(key, nonce, aad, bytes(bad))[0] == 0x1A
    # ... and POLYVAL by the levels (what a text beyond one round of chunk workgroups takes), reached by the switch
    try:
        L.uaes_debug_plan_disable(1 << uaes.arrangement_id("siv.chunks"))
        for n in (40000, 3 * MIB + 5):
            assert uaes.plan("siv", n, 9)[0] == "siv.levels"
            data, aad = orc.splitmix(n, n), rnd.randbytes(9)
            want = orc.gcmsiv_encrypt(key, nonce, aad, data)
            assert uaes.GCM_SIV_encrypt(key, nonce, aad, data) == want, ("siv.levels", n)
            assert uaes.GCM_SIV_decrypt(key, nonce, aad, want) == (0, data)
    finally:
        L.uaes_debug_plan_disable(0)


def test_two_launch_forms_without_a_counter_word(orc):
    """UAES_GCM_FOLD=0: no counter word is ever handed to the kernels, the one-launch arrangements take
    their two-launch form (chunk workgroups, then k_gcm_combine; GCM-SIV: the levels) -- in a process of its own, a few
    sizes of every such arrangement against the oracle"""
    import subprocess
    import sys
    code = r"""
import sys
sys.path.insert(0, %r)
import micro_aes_amd as uaes
from oracle.pyoracle import Oracle
orc = Oracle()
key, n12 = bytes(range(16)), bytes(range(12))
for n, aad in ((40000, b"abc"), ((1 << 20) + 5, b""), ((6 << 20) + 16, bytes(100)), ((20 << 20) + 3, b"x")):
    d = orc.splitmix(n + 1, n)
    want = orc.gcm_encrypt(key, n12, aad, d)
    assert uaes.AES_GCM_encrypt(key, n12, aad, d) == want, n
    assert uaes.AES_GCM_decrypt(key, n12, aad, want) == (0, d), n
    bad = bytearray(want); bad[len(bad) // 2] ^= 1
    assert uaes.AES_GCM_decrypt(key, n12, aad, bytes(bad))[0] == 0x1A
    ws = orc.gcmsiv_encrypt(key, n12, aad, d)
    assert uaes.GCM_SIV_encrypt(key, n12, aad, d) == ws, n
    assert uaes.GCM_SIV_decrypt(key, n12, aad, ws) == (0, d), n
s = uaes.GcmStream(key, n12, b"hdr")
d = orc.splitmix(3, (3 << 20) + 7)
out = s.update(d[: 1 << 20]) + s.update(d[1 << 20: 3 << 20]) + s.update(d[3 << 20:])
assert out + s.finish() == orc.gcm_encrypt(key, n12, b"hdr", d)
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, UAES_GCM_FOLD="0"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.split()[-1] == "ok", r.stderr[-2000:]
