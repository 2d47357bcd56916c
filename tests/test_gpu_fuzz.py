"""Seeded differential fuzzing of the C ABI against the CPU oracle: random key sizes,
sizes drawn around the kernels' internal boundaries (16 B blocks, 256-block chunks, 16-chunk
runs, 256 KiB CTR chunks and whole rounds of them, GHASH level plans), host or device
pointers, aligned or not, in place or not.  Every case is reproducible from its printed tuple."""
import ctypes as C
import os
import random

import pytest

import micro_aes_amd as uaes

pytestmark = pytest.mark.gpu

EDGES = [0, 1, 15, 16, 17, 31, 32, 48, 255, 256, 4095, 4096, 4097, 65535, 65536, 65537,
         256 * 16, 256 * 16 * 16, 256 * 16 * 16 + 16, 1 << 18, (1 << 18) + 16, 3 << 18, 1 << 20]


def pick_size(rnd, cap):
    r = rnd.random()
    if r < 0.5:
        n = rnd.choice(EDGES) + rnd.choice([0, 0, 0, 1, 16, -1, -16, 5])
    elif r < 0.8:
        n = rnd.randrange(0, 70000)
    else:
        n = rnd.randrange(0, cap)
    return max(0, min(n, cap))


class Buffers:
    """input/output placement for one case: host or device memory, offset from a 16-byte
    boundary or not, output aliased with the input or not"""

    def __init__(self, rnd, data, out_len):
        import torch
        self.torch = torch
        self.n_in, self.n_out = len(data), out_len
        self.dev_in, self.dev_out = rnd.random() < 0.5, rnd.random() < 0.5
        self.off_in, self.off_out = rnd.choice([0, 0, 1, 3, 8]), rnd.choice([0, 0, 2, 5, 8])
        self.alias = rnd.random() < 0.3
        cap = max(len(data), out_len) + 64
        self.guard = 0xA5
        if self.alias:
            self.dev_out, self.off_out = self.dev_in, self.off_in
        if self.dev_in:
            self.tin = torch.full((cap,), self.guard, dtype=torch.uint8, device="cuda:0")
            if data:
                self.tin[self.off_in:self.off_in + len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
            self.pin = C.c_void_p(self.tin.data_ptr() + self.off_in)
        else:
            self.hin = self._host(cap)
            C.memmove(C.addressof(self.hin) + self.off_in, data, len(data))
            self.pin = C.c_void_p(C.addressof(self.hin) + self.off_in)
        if self.alias:
            self.pout = self.pin
        elif self.dev_out:
            self.tout = torch.full((cap,), self.guard, dtype=torch.uint8, device="cuda:0")
            self.pout = C.c_void_p(self.tout.data_ptr() + self.off_out)
        else:
            self.hout = self._host(cap)
            self.pout = C.c_void_p(C.addressof(self.hout) + self.off_out)

    def _host(self, cap):
        b = (C.c_uint8 * cap)()
        C.memset(b, self.guard, cap)
        return b

    def result(self, n=None):
        """(output bytes, guard bytes behind them intact?)"""
        n = self.n_out if n is None else n
        if self.alias:
            src, off, dev = (self.tin if self.dev_in else self.hin), self.off_in, self.dev_in
        else:
            src, off, dev = (self.tout if self.dev_out else self.hout), self.off_out, self.dev_out
        if dev:
            self.torch.cuda.synchronize()
            raw = bytes(src.cpu().numpy())
        else:
            raw = bytes(src)
        tail = raw[off + max(n, self.n_in if self.alias else 0):]
        return raw[off:off + n], all(b == self.guard for b in tail[:16])

    def describe(self):
        return dict(dev_in=self.dev_in, dev_out=self.dev_out, off_in=self.off_in, off_out=self.off_out, alias=self.alias)


def run_cases(seed, count, body):
    # UAES_FUZZ_SEED shifts every seed: `for s in 1 2 3; do UAES_FUZZ_SEED=$s pytest tests/test_gpu_fuzz.py; done`
    rnd = random.Random(seed + 1000003 * int(os.environ.get("UAES_FUZZ_SEED", "0")))
    for i in range(count):
        body(rnd, i)


def test_fuzz_ctr_ecb(orc):
    L = uaes.engine()

    def body(rnd, i):
        bits = rnd.choice([128, 192, 256])
        key = rnd.randbytes(bits // 8)
        n = pick_size(rnd, 3 << 20)
        data = orc.splitmix(1000 + i, n)
        # CTR with a random start counter (56-bit carries) and block offset
        ctr0 = rnd.randbytes(9) + rnd.choice([bytes(7), b"\xff" * 7, rnd.randbytes(7), b"\0\0\0\xff\xff\xff\xfe"])
        off = rnd.choice([0, 1, 255, 256, 1 << 20, (1 << 32) - 3])
        b = Buffers(rnd, data, n)
        info = ("ctr", bits, n, off, ctr0.hex(), b.describe())
        assert L.uaes_ctr_xcrypt_at(bits, key, ctr0, off, b.pin, n, b.pout) == 0, info
        got, guard_ok = b.result()
        assert got == orc.ctr_xcrypt_at(key, ctr0, off, data) and guard_ok, info
        # ECB: N1 zero padding on encrypt, 0x1D on a ragged decrypt
        padded = (n + 15) // 16 * 16
        b = Buffers(rnd, data, padded)
        info = ("ecb", bits, n, b.describe())
        assert L.uaes_ecb_encrypt(bits, key, b.pin, n, b.pout) == 0, info
        got, guard_ok = b.result()
        want = orc.ecb_encrypt(key, data)
        assert got == want and guard_ok, info
        b = Buffers(rnd, want, padded)
        assert L.uaes_ecb_decrypt(bits, key, b.pin, padded, b.pout) == 0, info
        assert b.result()[0][:n] == data, info

    run_cases(101, 150, body)


def test_fuzz_xts(orc):
    L = uaes.engine()

    def body(rnd, i):
        bits = rnd.choice([128, 192, 256])
        keys, tweak = rnd.randbytes(bits // 4), rnd.randbytes(16)
        n = max(16, pick_size(rnd, 2 << 20))
        data = orc.splitmix(2000 + i, n)
        b = Buffers(rnd, data, n)
        info = ("xts", bits, n, b.describe())
        assert L.uaes_xts_encrypt(bits, keys, tweak, b.pin, n, b.pout) == 0, info
        got, guard_ok = b.result()
        rc, want = orc.xts(keys, tweak, data, True)
        assert rc == 0 and got == want and guard_ok, info
        b = Buffers(rnd, want, n)
        assert L.uaes_xts_decrypt(bits, keys, tweak, b.pin, n, b.pout) == 0, info
        assert b.result()[0] == data, info
        # batched data units of a random size (ciphertext stealing when ragged)
        sb = rnd.choice([16, 17, 31, 512, 520, 4096, 4099, 65536 + 7])
        ns = rnd.randrange(1, 40 if sb < 5000 else 4)
        if rnd.random() < 0.2:                       # units shorter than a chunk, beyond the one-launch limit: packed chunks
            sb = rnd.choice([64, 128, 192, 512, 1024, 1984, 2048, 4032])
            ns = rnd.randrange(4097, 4600) if sb >= 1024 else rnd.randrange(4097, 12000)
        first = rnd.choice([0, 1, (1 << 32) - 1, rnd.getrandbits(60)])
        data = orc.splitmix(2500 + i, sb * ns)
        b = Buffers(rnd, data, sb * ns)
        info = ("xts_sectors", bits, sb, ns, first, b.describe())
        assert L.uaes_xts_sectors(bits, keys, first, sb, ns, b.pin, b.pout, 1) == 0, info
        got, guard_ok = b.result()
        assert got == orc.xts_sectors(keys, first, sb, data, True)[1] and guard_ok, info

    run_cases(202, 100, body)


def test_fuzz_aead(orc):
    L = uaes.engine()
    modes = {
        "gcm": (L.uaes_gcm_encrypt, L.uaes_gcm_decrypt, orc.gcm_encrypt, 300 << 10),
        "ocb": (L.uaes_ocb_encrypt, L.uaes_ocb_decrypt, orc.ocb_encrypt, 2 << 20),
        "gcmsiv": (L.uaes_gcmsiv_encrypt, L.uaes_gcmsiv_decrypt, orc.gcmsiv_encrypt, 600 << 10),     # (long: uaesk_gcmsiv_long)
    }

    def body(rnd, i):
        name = rnd.choice(list(modes))
        enc, dec, ref, cap = modes[name]
        bits = rnd.choice([128, 192, 256])
        key, nonce = rnd.randbytes(bits // 8), rnd.randbytes(12)
        aad = rnd.randbytes(rnd.choice([0, 0, 1, 15, 16, 17, 100, 4096, 5000, 16385, 131072 + 5, 262144]))
        n = pick_size(rnd, cap)
        data = orc.splitmix(3000 + i, n)
        want = ref(key, nonce, aad, data)
        aad_arg = aad
        if aad and rnd.random() < 0.4:               # associated data in device memory, oddly aligned
            import torch
            off = rnd.choice([0, 1, 7])
            taad = torch.zeros(len(aad) + 16, dtype=torch.uint8, device="cuda:0")
            taad[off:off + len(aad)] = torch.frombuffer(bytearray(aad), dtype=torch.uint8).to("cuda:0")
            aad_arg = C.c_void_p(taad.data_ptr() + off)
        b = Buffers(rnd, data, n + 16)
        info = (name, bits, n, len(aad), not isinstance(aad_arg, bytes), b.describe())
        assert enc(bits, key, nonce, aad_arg, len(aad), b.pin, n, b.pout) == 0, info
        got, guard_ok = b.result()
        assert got == want and guard_ok, info
        b = Buffers(rnd, want, n)
        assert dec(bits, key, nonce, aad_arg, len(aad), b.pin, n, b.pout) == 0, info
        assert b.result()[0] == data, info
        # one flipped bit anywhere (text, tag or aad) must be rejected
        bad = bytearray(want)
        where = rnd.randrange(len(bad))
        bad[where] ^= 1 << rnd.randrange(8)
        b = Buffers(rnd, bytes(bad), n)
        assert dec(bits, key, nonce, aad, len(aad), b.pin, n, b.pout) == 0x1A, info + (where,)
        if name == "gcm" and not b.alias:            # N7: plaintext buffer untouched
            assert all(x == b.guard for x in b.result()[0]), info

    run_cases(303, 160, body)


def test_fuzz_feedback_and_macs(orc):
    L = uaes.engine()

    def body(rnd, i):
        bits = rnd.choice([128, 192, 256])
        key, iv = rnd.randbytes(bits // 8), rnd.randbytes(16)
        n = pick_size(rnd, 40 << 10)
        data = orc.splitmix(4000 + i, n)
        # CFB / OFB: any length
        for name, f_enc, f_dec, want in (
                ("cfb", L.uaes_cfb_encrypt, L.uaes_cfb_decrypt, orc.cfb(key, iv, data, True)),
                ("ofb", L.uaes_ofb_xcrypt, L.uaes_ofb_xcrypt, orc.ofb(key, iv, data))):
            b = Buffers(rnd, data, n)
            info = (name, bits, n, b.describe())
            assert f_enc(bits, key, iv, b.pin, n, b.pout) == 0, info
            got, guard_ok = b.result()
            assert got == want and guard_ok, info
            b = Buffers(rnd, want, n)
            assert f_dec(bits, key, iv, b.pin, n, b.pout) == 0, info
            assert b.result()[0] == data, info
        # CBC with CS3 stealing: len >= 16
        if n >= 16:
            rc, want = orc.cbc(key, iv, data, True)
            b = Buffers(rnd, data, n)
            info = ("cbc", bits, n, b.describe())
            assert L.uaes_cbc_encrypt(bits, key, iv, b.pin, n, b.pout) == 0 and rc == 0, info
            got, guard_ok = b.result()
            assert got == want and guard_ok, info
            b = Buffers(rnd, want, n)
            assert L.uaes_cbc_decrypt(bits, key, iv, b.pin, n, b.pout) == 0, info
            assert b.result()[0] == data, info
        # CMAC
        mac = (C.c_uint8 * 16)()
        b = Buffers(rnd, data, 0)
        assert L.uaes_cmac(bits, key, b.pin, n, mac) == 0
        assert bytes(mac) == orc.cmac(key, data), ("cmac", bits, n, b.describe())

    run_cases(404, 80, body)


def test_fuzz_cbc_without_cts_and_ctr_constants(orc):
    """the CTS 0 build's CBC (padded last chunk, whole-block decrypt) and CTR with other CTR_IV_LENGTH /
    CTR_START_VALUE, any placement of the buffers (host / device, misaligned, in place)"""
    L = uaes.engine()

    def body(rnd, i):
        bits = rnd.choice([128, 192, 256])
        key, iv = rnd.randbytes(bits // 8), rnd.randbytes(16)
        n = pick_size(rnd, 24 << 10)
        padding = rnd.choice([0, 1, 2])
        data = orc.splitmix(9000 + i, n)
        rc, want = orc.cbc_nocts(key, iv, data, True, padding=padding)
        b = Buffers(rnd, data, len(want))
        info = ("cbc-nocts", bits, n, padding, b.describe())
        assert rc == 0 and L.uaes_cbc_encrypt_padded(bits, key, iv, padding, b.pin, n, b.pout) == 0, info
        got, guard_ok = b.result()
        assert got == want and guard_ok, info
        b = Buffers(rnd, want, len(want))
        assert L.uaes_cbc_decrypt_blocks(bits, key, iv, b.pin, len(want), b.pout) == 0, info
        got, guard_ok = b.result()
        assert got == orc.cbc_nocts(key, iv, want, False)[1] and got[:n] == data and guard_ok, info
        if len(want) > 16:                                   # a ragged length is refused, nothing written
            b = Buffers(rnd, want, len(want))
            before = b.result()[0]
            assert L.uaes_cbc_decrypt_blocks(bits, key, iv, b.pin, len(want) - 3, b.pout) == 1, info
            assert b.result()[0] == before, info
        ivl, start = rnd.choice([0, 1, 8, 11, 12, 13, 16]), rnd.choice([0, 1, 2, 255, 256, 0xFFFFFFFF, rnd.getrandbits(64)])
        civ = rnd.randbytes(ivl)
        b = Buffers(rnd, data, n)
        info = ("ctr-iv", bits, n, ivl, start, b.describe())
        assert L.uaes_ctr_xcrypt_iv(bits, key, civ, ivl, start, b.pin, n, b.pout) == 0, info
        got, guard_ok = b.result()
        assert got == orc.ctr_encrypt_iv(key, civ, start, data) and guard_ok, info

    run_cases(505, 60, body)


def test_fuzz_nonce_and_tag_lengths(orc):
    """GCM / CCM / OCB with the reference's other compile-time lengths (micro_aes.h:103-116) as run-time arguments:
    random legal nonce and tag lengths, host or device buffers at odd offsets, in place or not; nothing is written
    behind the (possibly truncated) tag, a forged tag gives 0x1A and -- for GCM -- an untouched output (N7)"""
    L = uaes.engine()

    def body(rnd, i):
        bits = rnd.choice([128, 192, 256])
        key = rnd.randbytes(bits // 8)
        mode = rnd.choice(["gcm", "ccm", "ocb"])
        if mode == "gcm":
            nl, tl, cap = rnd.choice([12, 12, 1, 8, 13, 60]), rnd.randrange(1, 17), 3 << 20
        elif mode == "ccm":
            nl, tl, cap = rnd.randrange(7, 14), 2 * rnd.randrange(2, 9), 24 << 10       # the MAC is a serial chain
        else:
            nl, tl, cap = rnd.randrange(1, 16), rnd.randrange(1, 17), 300 << 10
        n = pick_size(rnd, cap)
        nonce, aad = rnd.randbytes(nl), rnd.randbytes(rnd.choice([0, 0, 1, 16, 33, 700]))
        pt = orc.splitmix(9000 + i, n)
        want = getattr(orc, mode + "_encrypt")(key, nonce, aad, pt, tag_len=tl)
        enc, dec = getattr(L, "uaes_%s_encrypt_ex" % mode), getattr(L, "uaes_%s_decrypt_ex" % mode)
        b = Buffers(rnd, pt, n + tl)
        info = (mode, bits, nl, tl, n, len(aad), b.describe())
        assert enc(bits, key, nonce, nl, tl, aad, len(aad), b.pin, n, b.pout) == 0, info
        got, guard_ok = b.result()
        assert got == want and guard_ok, info
        b = Buffers(rnd, want, n)
        assert dec(bits, key, nonce, nl, tl, aad, len(aad), b.pin, n, b.pout) == 0, info
        got, guard_ok = b.result()
        assert got == pt and guard_ok, info
        bad = bytearray(want)
        bad[n + rnd.randrange(tl)] ^= 1 << rnd.randrange(8)
        b = Buffers(rnd, bytes(bad), n)
        assert dec(bits, key, nonce, nl, tl, aad, len(aad), b.pin, n, b.pout) == 0x1A, info
        if mode == "gcm" and not b.alias:
            assert all(x == b.guard for x in b.result()[0]), info

    run_cases(606, 120, body)


def test_fuzz_ctr_large_sizes_piecewise(orc):
    """60-200 MiB texts (several rounds of 256 KiB chunks plus thin last rounds): one call equals
    three calls over random 16-byte-aligned cuts with the block offset advanced, and the first and
    last 64 KiB equal the oracle's"""
    import torch
    rnd = random.Random(505 + 1000003 * int(os.environ.get("UAES_FUZZ_SEED", "0")))
    for _ in range(5):
        bits = rnd.choice([128, 256])
        key = rnd.randbytes(bits // 8)
        n = rnd.randrange(60 << 20, 200 << 20) // 16 * 16 + rnd.choice([0, 0, 7])
        ctr0 = rnd.randbytes(12) + b"\xff\xff" + rnd.randbytes(2)
        src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0")
        one = torch.empty_like(src)
        uaes.ctr_xcrypt_dev(key, ctr0, 5, src, one, nbytes=n)
        cuts = sorted(rnd.randrange(0, n // 16) * 16 for _ in range(2))
        parts = torch.empty_like(src)
        for lo, hi in zip([0] + cuts, cuts + [n]):
            if hi > lo:
                uaes.ctr_xcrypt_dev(key, ctr0, 5 + lo // 16, src[lo:hi], parts[lo:hi], nbytes=hi - lo)
        torch.cuda.synchronize()
        assert torch.equal(one, parts), (bits, n, cuts)
        m = 1 << 16
        head = bytes(src[:m].cpu().numpy())
        assert bytes(one[:m].cpu().numpy()) == orc.ctr_xcrypt_at(key, ctr0, 5, head)
        t0 = (n - m) // 16 * 16
        tail = bytes(src[t0:].cpu().numpy())
        assert bytes(one[t0:].cpu().numpy()) == orc.ctr_xcrypt_at(key, ctr0, 5 + t0 // 16, tail)


def test_fuzz_gcm_one_pass_sizes(orc):
    """8-70 MiB GCM encryptions take the one-pass kernel: random sizes (stripe counts that do not
    divide by the grid, ragged tails), random AAD up to 2 MiB, device buffers in and out of place.
    Ciphertext == the CTR path from J0+1 (itself pinned to the oracle), the tag is accepted by the
    two-pass decrypt (separate GHASH levels) and by the one-pass decrypt, and rejected by both after a
    bit flip (output untouched / zeroed); the shortest case of a run is also checked against the oracle
    end to end."""
    import torch
    rnd = random.Random(606 + 1000003 * int(os.environ.get("UAES_FUZZ_SEED", "0")))
    cases = sorted((rnd.randrange(8 << 20, 70 << 20) + rnd.choice([0, 0, 1, 15, 16, 4064, 4080]) for _ in range(6)))
    for i, n in enumerate(cases):
        bits = rnd.choice([128, 192, 256])
        key, nonce = rnd.randbytes(bits // 8), rnd.randbytes(12)
        alen = rnd.choice([0, 0, 7, 16, 4096, 65536 + 3, rnd.randrange(0, 2 << 20)])
        aad = rnd.randbytes(alen)
        a = torch.frombuffer(bytearray(aad), dtype=torch.uint8).to("cuda:0") if alen else None
        src = torch.randint(0, 256, (n + 16,), dtype=torch.uint8, device="cuda:0")
        inplace = rnd.random() < 0.4
        dst = src.clone() if inplace else torch.full((n + 32,), 0xA5, dtype=torch.uint8, device="cuda:0")
        info = (bits, n, alen, inplace)
        uaes.gcm_encrypt_dev(key, nonce, a, dst if inplace else src, n, dst)
        ref = torch.empty(n, dtype=torch.uint8, device="cuda:0")
        uaes.ctr_xcrypt_dev(key, nonce + b"\0\0\0\1", 1, src, ref, nbytes=n)       # keystream block i uses J0 + 1 + i (N4)
        torch.cuda.synchronize()
        assert torch.equal(dst[:n], ref), info
        if not inplace:
            assert int((dst[n + 16:] != 0xA5).sum()) == 0, info
        status = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
        back = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
        uaes.gcm_decrypt_dev(key, nonce, a, dst, n, back, status)
        torch.cuda.synchronize()
        assert int(status.item()) == 0 and torch.equal(back, src[:n]), info
        # the one-pass decrypt (uaes_set_gcm_one_pass_decrypt: the caller accepts a wiped buffer on failure): same plaintext,
        # out of place and in place
        L = uaes.engine()
        try:
            L.uaes_set_gcm_one_pass_decrypt(1)
            back.fill_(0x5A)
            status.fill_(-1)
            uaes.gcm_decrypt_dev(key, nonce, a, dst, n, back, status)
            torch.cuda.synchronize()
            assert int(status.item()) == 0 and torch.equal(back, src[:n]), info
            work = dst[: n + 16].clone()
            uaes.gcm_decrypt_dev(key, nonce, a, work, n, work, status)
            torch.cuda.synchronize()
            assert int(status.item()) == 0 and torch.equal(work[:n], src[:n]) and torch.equal(work[n:], dst[n:n + 16]), info
            dst[rnd.randrange(n + 16)] ^= 1 << rnd.randrange(8)
            back.fill_(0x5A)
            uaes.gcm_decrypt_dev(key, nonce, a, dst, n, back, status)
            torch.cuda.synchronize()
            assert int(status.item()) == 0x1A and int(back.sum()) == 0, info         # forged: zeroed
        finally:
            L.uaes_set_gcm_one_pass_decrypt(0)
        back.fill_(0x5A)
        uaes.gcm_decrypt_dev(key, nonce, a, dst, n, back, status)
        torch.cuda.synchronize()
        assert int(status.item()) == 0x1A and int((back != 0x5A).sum()) == 0, info    # forged, default: untouched
        if i == 0:
            pt = bytes(src[:n].cpu().numpy())
            got = uaes.AES_GCM_encrypt(key, nonce, aad, pt)
            want = orc.gcm_encrypt(key, nonce, aad, pt)
            assert got[-16:] == want[-16:] and got[:4096] == want[:4096] and got[-4096:] == want[-4096:], info


def test_fuzz_gcm_record_calls(orc):
    """uaes_gcm_key_{en,de}crypt_records against the oracle's AES_GCM_encrypt of every record by itself: random record
    counts (more and fewer than one workgroup turn holds), record and AAD lengths across the kernel's arrangements,
    shared or per-record AAD, strides with gaps, host or device buffers, in place; forged records keep the output's
    guard bytes (N7) while their neighbours decrypt."""
    import torch

    def body(rnd, i):
        bits = rnd.choice([128, 192, 256])
        key = rnd.randbytes(bits // 8)
        k = uaes.GcmKey(key)
        try:
            aad_len = rnd.choice([0, 0, 5, 13, 16, 17, rnd.randrange(0, 200)])
            cap = k.record_max(aad_len)
            rec_len = rnd.choice([rnd.randrange(0, 64), rnd.randrange(0, 1100), rnd.randrange(900, 2100), rnd.randrange(2000, 4200),
                                  rnd.randrange(4000, 8300), rnd.randrange(8000, 16500), rnd.randrange(16000, cap + 1), cap,
                                  16 * rnd.randrange(0, 64)])
            nrec = rnd.choice([1, 2, rnd.randrange(1, 40), rnd.randrange(1, 40), rnd.randrange(200, 700)])
            if rec_len > 9000:
                nrec = min(nrec, 24)                        # the oracle takes ~1 ms per 16 KiB
            stride = (rec_len + 16 + 15) // 16 * 16 + 16 * rnd.choice([0, 0, 1, 3, 16])
            per_record_aad = aad_len > 0 and rnd.random() < 0.5
            aad_stride = (aad_len + rnd.choice([0, 1, 7])) if per_record_aad else 0
            nonces = rnd.randbytes(12 * nrec)
            aad = rnd.randbytes(aad_stride * (nrec - 1) + aad_len if per_record_aad else aad_len)
            plain = bytearray(rnd.randbytes(stride * nrec))
            info = dict(seed_case=i, bits=bits, nrec=nrec, rec_len=rec_len, aad_len=aad_len, stride=stride, aad_stride=aad_stride)
            want = [orc.gcm_encrypt(key, nonces[12 * r: 12 * r + 12], aad[aad_stride * r: aad_stride * r + aad_len],
                                    bytes(plain[stride * r: stride * r + rec_len])) for r in range(nrec)]
            L = uaes.engine()
            u8 = lambda b: (C.c_uint8 * max(len(b), 1)).from_buffer_copy(bytes(b) if len(b) else b"\0")
            on_device = rnd.random() < 0.5
            if on_device:
                dev = lambda b: torch.frombuffer(bytearray(b) if len(b) else bytearray(1), dtype=torch.uint8).to("cuda:0")
                tn, ta, tp = dev(nonces), dev(aad), dev(plain)
                out = tp if rnd.random() < 0.4 else torch.full((stride * nrec,), 0xA5, dtype=torch.uint8, device="cuda:0")
                k.encrypt_records_dev(nrec, tn, ta if aad_len else None, aad_len, aad_stride, tp, rec_len, stride, out, stride)
                torch.cuda.synchronize()
                got = bytes(out.cpu().numpy())
            else:
                ob = (C.c_uint8 * (stride * nrec))()
                C.memset(ob, 0xA5, stride * nrec)
                rc = L.uaes_gcm_key_encrypt_records(k._h, nrec, u8(nonces), u8(aad) if aad_len else None, aad_len, aad_stride,
                                                    u8(plain), rec_len, stride, ob, stride)
                assert rc == 0, info
                got = bytes(ob)
            for r in range(nrec):
                assert got[stride * r: stride * r + rec_len + 16] == want[r], (info, r)
            # decrypt with some records forged
            forged = set(rnd.sample(range(nrec), rnd.choice([0, 0, 1, min(nrec, 3)])))
            ct = bytearray(stride * nrec)
            for r in range(nrec):
                rec = bytearray(want[r])
                if r in forged:
                    rec[rnd.randrange(len(rec))] ^= 1 << rnd.randrange(8)
                ct[stride * r: stride * r + rec_len + 16] = rec
            ob = (C.c_uint8 * (stride * nrec))()
            C.memset(ob, 0xA5, stride * nrec)
            ver = (C.c_uint8 * nrec)()
            rc = L.uaes_gcm_key_decrypt_records(k._h, nrec, u8(nonces), u8(aad) if aad_len else None, aad_len, aad_stride,
                                                u8(ct), rec_len, stride, ob, stride, ver)
            assert rc == (0x1A if forged else 0), info
            back = bytes(ob)
            for r in range(nrec):
                assert ver[r] == (0x1A if r in forged else 0), (info, r)
                keep = b"\xa5" * rec_len if r in forged else bytes(plain[stride * r: stride * r + rec_len])
                assert back[stride * r: stride * r + rec_len] == keep, (info, r)
                assert back[stride * r + rec_len: stride * (r + 1)] == b"\xa5" * (stride - rec_len), (info, r)
        finally:
            k.close()

    run_cases(909, 14, body)


def test_fuzz_gcm_record_calls_of_different_lengths(orc):
    """uaes_gcm_key_{en,de}crypt_records_v: random slot sizes, random lengths per record (zero and the maximum among them),
    shared or per-record AAD, some records forged -- every record against the oracle's call of that record alone"""
    def body(rnd, i):
        bits = rnd.choice([128, 192, 256])
        key = rnd.randbytes(bits // 8)
        k = uaes.GcmKey(key)
        try:
            aad_len = rnd.choice([0, 0, 5, 13, 16, rnd.randrange(0, 100)])
            cap = k.record_max(aad_len)
            max_len = rnd.choice([rnd.randrange(0, 64), rnd.randrange(0, 1100), rnd.randrange(900, 2100), rnd.randrange(2000, 8300),
                                  rnd.randrange(8000, cap + 1), cap])
            nrec = rnd.choice([1, 2, rnd.randrange(1, 40), rnd.randrange(100, 400)])
            if max_len > 6000:
                nrec = min(nrec, 20)
            lens = [rnd.choice([0, max_len, rnd.randrange(0, max_len + 1)]) for _ in range(nrec)]
            per_record_aad = aad_len > 0 and rnd.random() < 0.5
            nonces = [rnd.randbytes(12) for _ in range(nrec)]
            aads = [rnd.randbytes(aad_len) for _ in range(nrec)] if per_record_aad else rnd.randbytes(aad_len)
            aad_of = (lambda r: aads[r]) if per_record_aad else (lambda r: aads)
            recs = [rnd.randbytes(n) for n in lens]
            info = dict(seed_case=i, bits=bits, nrec=nrec, max_len=max_len, aad_len=aad_len, per_record_aad=per_record_aad)
            want = [orc.gcm_encrypt(key, nonces[r], aad_of(r), recs[r]) for r in range(nrec)]
            stride = (max_len + 16 + 15) // 16 * 16 + 16 * rnd.choice([0, 0, 2])
            got = k.encrypt_records_v(nonces, aads, recs, max_len=max_len, stride=stride)
            assert got == want, info
            forged = set(rnd.sample(range(nrec), rnd.choice([0, 1, min(nrec, 3)])))
            spoiled = list(got)
            for r in forged:
                b = bytearray(spoiled[r])
                b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
                spoiled[r] = bytes(b)
            rc, ver, texts = k.decrypt_records_v(nonces, aads, spoiled, prefill=0xC3, max_len=max_len, stride=stride)
            assert rc == (0x1A if forged else 0), info
            for r in range(nrec):
                assert ver[r] == (0x1A if r in forged else 0), (info, r)
                assert texts[r] == (b"\xc3" * lens[r] if r in forged else recs[r]), (info, r)
        finally:
            k.close()

    run_cases(1010, 12, body)


def test_fuzz_mgpu_split(orc):
    """uaes_mgpu_{ctr,ecb,gcm,xts}: random device lists over the visible devices (an ordinal may repeat), random
    buffer placement (host / device, aligned or not, in place), sizes around the slicing boundaries -- the result is the
    one-call result whatever the list; a forged GCM text leaves the plaintext buffer as it was (N7 across devices)"""
    import torch
    L = uaes.engine()
    ndev_visible = torch.cuda.device_count()

    def body(rnd, i):
        nd = rnd.choice([1, 2, 3, 5, 8, 16])
        devlist = [rnd.randrange(ndev_visible) for _ in range(nd)]

        def devs_for(b):
            # device buffers of this test live on cuda:0 and no peer access is set up between the devices of a box: a
            # case with device memory runs all its slices on device 0, host buffers go to the random list
            dl = [0] * nd if (b.dev_in or b.dev_out) else devlist
            return (C.c_int * nd)(*dl)
        bits = rnd.choice([128, 192, 256])
        key = rnd.randbytes(bits // 8)
        n = pick_size(rnd, 1 << 20) if rnd.random() < 0.8 else rnd.randrange(0, 16 * nd + 40)
        data = orc.splitmix(7000 + i, n)
        which = rnd.choice(["ctr", "ecb", "gcm", "xts"])
        if which == "ctr":
            ctr0, off = rnd.randbytes(9) + rnd.choice([bytes(7), b"\xff" * 7, rnd.randbytes(7)]), rnd.choice([0, 3, (1 << 40) + 7])
            b = Buffers(rnd, data, n)
            info = ("mgpu-ctr", devlist, bits, n, b.describe())
            assert L.uaes_mgpu_ctr_xcrypt_at(nd, devs_for(b), bits, key, ctr0, off, b.pin, n, b.pout) == 0, info
            got, guard_ok = b.result()
            assert got == orc.ctr_xcrypt_at(key, ctr0, off, data) and guard_ok, info
        elif which == "ecb":
            padding = rnd.choice([0, 0, 1, 2])
            want = orc.ecb_encrypt(key, data, padding)
            b = Buffers(rnd, data, len(want))
            info = ("mgpu-ecb", devlist, bits, n, padding, b.describe())
            assert L.uaes_mgpu_ecb_encrypt(nd, devs_for(b), bits, key, padding, b.pin, n, b.pout) == 0, info
            got, guard_ok = b.result()
            assert got == want and guard_ok, info
            b = Buffers(rnd, want, len(want))
            assert L.uaes_mgpu_ecb_decrypt(nd, devs_for(b), bits, key, b.pin, len(want), b.pout) == 0, info
            assert b.result()[0][:n] == data, info
        elif which == "gcm":
            nonce, aad = rnd.randbytes(12), rnd.randbytes(rnd.choice([0, 0, 1, 16, 33, 4097]))
            want = orc.gcm_encrypt(key, nonce, aad, data)
            b = Buffers(rnd, data, n + 16)
            info = ("mgpu-gcm", devlist, bits, n, len(aad), b.describe())
            assert L.uaes_mgpu_gcm_encrypt(nd, devs_for(b), bits, key, nonce, aad, len(aad), b.pin, n, b.pout) == 0, info
            got, guard_ok = b.result()
            assert got == want and guard_ok, info
            b = Buffers(rnd, want, n)
            assert L.uaes_mgpu_gcm_decrypt(nd, devs_for(b), bits, key, nonce, aad, len(aad), b.pin, n, b.pout) == 0, info
            assert b.result()[0] == data, info
            bad = bytearray(want)
            bad[rnd.randrange(len(bad))] ^= 1 << rnd.randrange(8)
            b = Buffers(rnd, bytes(bad), n)
            assert L.uaes_mgpu_gcm_decrypt(nd, devs_for(b), bits, key, nonce, aad, len(aad), b.pin, n, b.pout) == 0x1A, info
            if not b.alias:
                assert all(x == b.guard for x in b.result()[0]), info
        else:
            keys = rnd.randbytes(bits // 4)
            sb = rnd.choice([16, 17, 512, 4096, 4099])
            ns = rnd.randrange(1, 60)
            first = rnd.choice([0, 1, (1 << 32) - 1, rnd.getrandbits(60)])
            data = orc.splitmix(7500 + i, sb * ns)
            b = Buffers(rnd, data, sb * ns)
            info = ("mgpu-xts", devlist, bits, sb, ns, first, b.describe())
            assert L.uaes_mgpu_xts_sectors(nd, devs_for(b), bits, keys, first, sb, ns, b.pin, b.pout, 1) == 0, info
            got, guard_ok = b.result()
            assert got == orc.xts_sectors(keys, first, sb, data, True)[1] and guard_ok, info

    run_cases(707, 120, body)


def test_fuzz_kernels_against_the_host_path(orc):
    """the two product implementations against each other: every one-message mode on random inputs, once on the GPU
    (default policy) and once on the engine's own host path (forced for the second call and switched off again) -- and
    both against the oracle.  Also what the opt-in promises: device pointers never take the host path."""
    import torch
    L = uaes.engine()

    def both(fn):
        gpu = fn()
        prev = uaes.host_policy(1 << 62, 1, 0)
        try:
            host = fn()
        finally:
            uaes.host_policy(*prev)
        assert uaes.host_policy() == (0, 0, 0)
        return gpu, host

    def body(rnd, i):
        bits = rnd.choice([128, 192, 256])
        key, iv, nonce = rnd.randbytes(bits // 8), rnd.randbytes(16), rnd.randbytes(12)
        n = pick_size(rnd, 200000)
        data, aad = orc.splitmix(9000 + i, n), rnd.randbytes(rnd.choice([0, 1, 16, 40]))
        info = (bits, n, len(aad))
        g, h = both(lambda: uaes.AES_CTR_encrypt(key, nonce, data))
        assert g == h == orc.ctr_encrypt(key, nonce, data), ("ctr",) + info
        pad = rnd.choice([0, 1, 2])
        g, h = both(lambda: uaes.AES_ECB_encrypt(key, data, pad))
        assert g == h == orc.ecb_encrypt(key, data, pad), ("ecb",) + info
        g, h = both(lambda: uaes.AES_GCM_encrypt(key, nonce, aad, data))
        assert g == h == orc.gcm_encrypt(key, nonce, aad, data), ("gcm",) + info
        g2, h2 = both(lambda: uaes.AES_GCM_decrypt(key, nonce, aad, g))
        assert g2 == h2 == (0, data)
        g, h = both(lambda: uaes.AES_OCB_encrypt(key, nonce, aad, data))
        assert g == h == orc.ocb_encrypt(key, nonce, aad, data), ("ocb",) + info
        g, h = both(lambda: uaes.GCM_SIV_encrypt(key, nonce, aad, data[:30000]))
        assert g == h == orc.gcmsiv_encrypt(key, nonce, aad, data[:30000]), ("gcmsiv",) + info
        g, h = both(lambda: uaes.AES_CCM_encrypt(key, nonce[:11], aad, data[:20000]))
        assert g == h == orc.ccm_encrypt(key, nonce[:11], aad, data[:20000]), ("ccm",) + info
        g, h = both(lambda: uaes.AES_CMAC(key, data[:20000]))
        assert g == h == orc.cmac(key, data[:20000]), ("cmac",) + info
        g, h = both(lambda: uaes.AES_CFB_encrypt(key, iv, data[:20000]))
        assert g == h == orc.cfb(key, iv, data[:20000], True), ("cfb",) + info
        g, h = both(lambda: uaes.AES_OFB_encrypt(key, iv, data[:20000]))
        assert g == h == orc.ofb(key, iv, data[:20000]), ("ofb",) + info
        if n >= 16:
            keys = rnd.randbytes(bits // 4)
            g, h = both(lambda: uaes.AES_XTS_encrypt(keys, iv, data))
            assert g == h == orc.xts(keys, iv, data, True), ("xts",) + info
            g, h = both(lambda: uaes.AES_CBC_encrypt(key, iv, data[:20000 + n % 16]))
            assert g == h and g[1] == orc.cbc(key, iv, data[:20000 + n % 16], True)[1], ("cbc",) + info
            g, h = both(lambda: uaes.AES_CBC_decrypt(key, iv, g[1]))
            assert g == h == (0, data[:20000 + n % 16])

    run_cases(909, 40, body)
    # device pointers stay on the GPU even with the host path switched on: a host routine would fault on them
    prev = uaes.host_policy(1 << 62, 1, 0)
    try:
        t = torch.zeros(4096, dtype=torch.uint8, device="cuda:0")
        o = torch.zeros(4096 + 16, dtype=torch.uint8, device="cuda:0")
        assert L.uaes_ctr_xcrypt(128, bytes(16), bytes(12), C.c_void_p(t.data_ptr()), 4096, C.c_void_p(o.data_ptr())) == 0
        assert bytes(o[:4096].cpu().numpy()) == orc.ctr_encrypt(bytes(16), bytes(12), bytes(4096))
        assert L.uaes_cmac(128, bytes(16), C.c_void_p(t.data_ptr()), 4096, (C.c_uint8 * 16)()) == 0
    finally:
        uaes.host_policy(*prev)
