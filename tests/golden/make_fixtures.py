#!/usr/bin/env python3
"""Generate the committed parity fixtures under tests/golden/.

Run in the BUILD CONTAINER (needs /root/reference and `make -C oracle`):

    python tests/golden/make_fixtures.py [--big]

What it writes (all of it DATA -- inputs and expected outputs; no reference
source text):

* ``GcmEncryptExtIV{128,192,256}.rsp`` -- the NIST CAVP GCM files held by the
  reference's own harness (testvectors/), reduced to the sections that harness
  actually runs (IVlen == 96 and Taglen == 128, aes_testvectors_GCM.h:86) plus the IVlen 8 and 1024
  sections that builds with GCM_NONCE_LEN = 1 / 128 run, so the repo carries 3 x ~0.4 MB instead of
  3 x ~3 MB.  Format unchanged.
* ``XTSGenAES{128,256}.rsp``, ``CMACGenAES{128,192,256}.rsp``, ``VNT{128,192,256}.rsp``
  -- NIST CAVP XTS / CMAC / CCM files, unmodified; ``SIV_GCM_ACVP.tv`` -- the 102 ACVP
  AES-GCM-SIV vectors the reference's harness holds, unmodified; ``OCB_AES128.tv`` -- its
  OpenSSL OCB vectors, unmodified.
* ``main_kats.json`` -- the hot-path known answers of the reference's main.c
  (main.c:16-34,49-50,58-60), re-verified here against the compiled reference.
* ``ref_vectors.json`` -- outputs of the COMPILED REFERENCE (oracle/_ref) on
  seeded inputs over the edge-case lengths (0, 1, 15, 16, 17, ... 4097, 65541)
  for ECB/CTR/XTS/GCM at 128/192/256 bits, plus error-path behaviour (N1,N5,N7).
* ``ecb_padding_vectors.json`` -- AES_ECB_encrypt of reference builds with AES_PADDING 1 / 2
  (micro_aes.h:79) on seeded inputs, and the main.c AES-192 PKCS#7 known answer (main.c:86,139).
* ``build_variant_vectors.json`` -- AES_CBC_* of reference builds with CTS 0 (micro_aes.h:56) and each AES_PADDING,
  AES_CTR_encrypt of builds with other CTR_IV_LENGTH / CTR_START_VALUE (micro_aes.h:98-99), and main.c's CTS 0
  known answer (main.c:36-40); ``--variants`` rewrites only this file.
* ``digests.json`` -- SHA-256 digests of the reference's output on the
  BASELINE.json workloads.  Cheap ones are recomputed
  here; the multi-GiB ones are recomputed only with --big (minutes of CPU).
"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, Reference, build  # noqa: E402

REF_TV = "/root/reference/testvectors"


def filter_gcm(src, dst):
    """keep only the [Taglen = 128] sections with [IVlen = 96] (the default GCM_NONCE_LEN) and 8 / 1024
    (what builds with GCM_NONCE_LEN = 1 / 128 run: J0 = GHASH(nonce), micro_aes.c:1145-1149)"""
    out, keep, hdr = [], False, {}
    with open(src) as f:
        lines = f.read().splitlines()
    i = 0
    preamble_done = False
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("["):
            # a section header is a run of [...] lines
            j, hdr = i, {}
            while j < len(lines) and lines[j].startswith("["):
                k, v = lines[j].strip("[]").split("=")
                hdr[k.strip()] = int(v)
                j += 1
            keep = hdr.get("IVlen") in (8, 96, 1024) and hdr.get("Taglen") == 128
            if keep:
                out.extend(lines[i:j])
            preamble_done = True
            i = j
            continue
        if not preamble_done or keep:
            out.append(ln)
        i += 1
    with open(dst, "w") as f:
        f.write("\n".join(out) + "\n")


def splitmix(orc, seed, n):
    return orc.splitmix(seed, (n + 7) // 8 * 8)[:n]


def enc_out(b):
    """full hex when small, digest + ends otherwise"""
    if len(b) <= 160:
        return {"hex": b.hex()}
    return {"len": len(b), "sha256": hashlib.sha256(b).hexdigest(),
            "head": b[:16].hex(), "tail": b[-16:].hex()}


MAIN_PT = ("c9f775baafa36c25cd610d3c75a482eadda97ca4864cdfe06eaf70a0ec0d7191"
           "d55027cf8f900214e634412583ff0b478EA2B7CA516745BFEA")
MAIN_IV = "8EA2B7CA516745BFEAfc49904b496089"
MAIN_KEY = ("279fb74a7572135e8f9b8ef6d1eee00369c4e0d86a7b0430d8cdb78070b4c55a"
            "00112233445566778899AABBCCDDEEFF000102030405060708090A0B0C0D0E0F")
MAIN_AUTH = "000102030405060708090A0B0C0D0E0F101112131415161718191A1B1C1D1E1F"
MAIN_KATS = {
    "ecb128": "5d00c273f8b2607da834632dcbb521f4697dd4ab20bb064532a6545e24e33ae9"
              "f545176111f93773dbecd262841cf83b10d145e71b772cf7a12889cda84be795",
    "ctr128": "6c6bae886c235d8c7997d45c1bf0bca248b4bca9eb396d1bf6945e5b7a4fc10f"
              "488cfe76fd5eaeff2b8fb469f78fa61e285e4cf9b9aee3d0a8",
    "xts128": "10f9301a157bfceb3eb9e7bd38500b7e959e21ba3cc1179ad7f7d7d99460e695"
              "5e8bcb177571c7196de58ff28c381913e7c82d0adfd90c45ca",
    "xts256": "40bfcc14845b1bb415dd13abf1e6f89d3bfd794cf6655ffd14c0d7e4177eeaf4"
              "5dd95f05663fcfb447671154a91b9d00d1bd7a35c14c74109a",
    "gcm128": "5ceab5b7c2d6dede555a23c7e3e632744075a51df482730ba31485ec987ddcc8"
              "73acdcfc6759a47ba424d838e7c0cb71b9a4d8f4572e214118c8ab284ca845c1"
              "4394618703cddf3afb",
    "cbc128": "65c48fdf9fbd626128f2d8bac3f7125175e7f4821fda026370011632779d7403"
              "c119ef461ac4e1bc8a7e36bf92b3b3d17E9E2D298E154BC42D",
    "cfb128": "edab3105e673bc9eb9102539a9f457bc245c14e1bff81b5b4a4a147c988cb0a6"
              "3f9c56525efbe64a876ad1d761d3fc9359fb4f5b2354acd490",
    "ofb128": "edab3105e673bc9eb9102539a9f457bcd28c8e4c92995f5cd9426926be1e775d"
              "e22b8ce4d0278b18181b8bec93b9726f959aa5d701d46102f0",
    "gcmsiv128": "2f1488496ada3f709760420ac72e5acfa977f6add4c55ac685f1b9dff8f381e0"
                 "2a64bbdd64cdd778525462949bb0b141db908c5cfa3657503666f879ac879fcb"
                 "f25c15d496a1e6f7f8",
    "ocb128": "fc254896eb785b05dd87f240722dd93561f5a0ef6aff2eb65953da0b26257ed0"
              "d69cb496e9a0cb1bf646151aa07e629a28d99f0ffd7ea7535c39f440df33c988"
              "c55cbcc8ac086ffa23",
    "cmac128": "b887df1fd8c239c3e8a64d9822e21128",
    "ccm128": "d2575123438338d70b2955537fdfcf41729870884e85af15f0a74975a72b337d"
              "04d426de87594b9abe3e6dcf07f21c99db3999f81299d302ad1e5ba683e9039a"
              "5483685f1bd2c3fa3b",
    "gcm256": "eb0f39c8cc86af343545fec3abc4d1fd26241218546289ec5ce5208e01873e90"
              "e86772931b80d74922565b38d35fe11a387b347949dda0879ca5f20fc9357760"
              "4b2f659e3b1d1b0f33",
}


def main_kats():
    """main.c constants as a table; verified against the compiled reference."""
    pt, iv = bytes.fromhex(MAIN_PT), bytes.fromhex(MAIN_IV)
    key = bytes.fromhex(MAIN_KEY)
    aad = bytes.fromhex(MAIN_AUTH)[1:]           # main.c:118  a = authKey + 1
    r128, r256 = Reference(128), Reference(256)
    kats = []

    def add(name, mode, bits, k, extra, out):
        kats.append(dict(name=name, mode=mode, keybits=bits, key=k.hex(),
                         pt=pt.hex(), expect=out.lower(), **extra))

    assert r128.ecb_encrypt(key[:16], pt).hex() == MAIN_KATS["ecb128"].lower()
    add("main.c:139 ECB", "ecb", 128, key[:16], {}, MAIN_KATS["ecb128"])
    assert r128.ctr_encrypt(key[:16], iv, pt).hex() == MAIN_KATS["ctr128"].lower()
    add("main.c:167 CTR", "ctr", 128, key[:16], {"iv": iv[:12].hex()}, MAIN_KATS["ctr128"])
    assert r128.xts(key[:32], iv, pt)[1].hex() == MAIN_KATS["xts128"].lower()
    add("main.c:174 XTS-128", "xts", 128, key[:32], {"tweak": iv.hex()}, MAIN_KATS["xts128"])
    assert r256.xts(key[:64], iv, pt)[1].hex() == MAIN_KATS["xts256"].lower()
    add("main.c:174 XTS-256", "xts", 256, key[:64], {"tweak": iv.hex()}, MAIN_KATS["xts256"])
    assert r128.gcm_encrypt(key[:16], iv, aad, pt).hex() == MAIN_KATS["gcm128"].lower()
    add("main.c:191 GCM-128", "gcm", 128, key[:16], {"nonce": iv[:12].hex(), "aad": aad.hex()},
        MAIN_KATS["gcm128"])
    assert r256.gcm_encrypt(key[:32], iv, aad, pt).hex() == MAIN_KATS["gcm256"].lower()
    add("main.c:191 GCM-256", "gcm", 256, key[:32], {"nonce": iv[:12].hex(), "aad": aad.hex()},
        MAIN_KATS["gcm256"])
    assert r128.cbc(key[:16], iv, pt)[1].hex() == MAIN_KATS["cbc128"].lower()
    add("main.c:146 CBC (CTS)", "cbc", 128, key[:16], {"iv": iv.hex()}, MAIN_KATS["cbc128"])
    assert r128.cfb(key[:16], iv, pt).hex() == MAIN_KATS["cfb128"]
    add("main.c:153 CFB", "cfb", 128, key[:16], {"iv": iv.hex()}, MAIN_KATS["cfb128"])
    assert r128.ofb(key[:16], iv, pt).hex() == MAIN_KATS["ofb128"]
    add("main.c:160 OFB", "ofb", 128, key[:16], {"iv": iv.hex()}, MAIN_KATS["ofb128"])
    assert r128.gcmsiv_encrypt(key[:16], iv[:12], aad, pt).hex() == MAIN_KATS["gcmsiv128"]
    add("main.c:219 GCM-SIV", "gcmsiv", 128, key[:16], {"nonce": iv[:12].hex(), "aad": aad.hex()}, MAIN_KATS["gcmsiv128"])
    assert r128.ocb_encrypt(key[:16], iv[:12], aad, pt).hex() == MAIN_KATS["ocb128"]
    add("main.c:205 OCB", "ocb", 128, key[:16], {"nonce": iv[:12].hex(), "aad": aad.hex()}, MAIN_KATS["ocb128"])
    # the RFC 7253 vector of main.c:262-273 (own plaintext)
    o_k, o_n = bytes.fromhex("000102030405060708090A0B0C0D0E0F"), bytes.fromhex("BBAA99887766554433221107")
    o_a = o_p = bytes.fromhex("000102030405060708090A0B0C0D0E0F1011121314151617")
    o_c = "1ca2207308c87c010756104d8840ce1952f09673a448a122c92c62241051f57356d7f3c90bb0e07f"
    assert r128.ocb_encrypt(o_k, o_n, o_a, o_p).hex() == o_c
    kats.append(dict(name="main.c:264 RFC-7253", mode="ocb", keybits=128, key=o_k.hex(), pt=o_p.hex(),
                     nonce=o_n.hex(), aad=o_a.hex(), expect=o_c))
    # the two RFC 8452 vectors of main.c:275-297 (own plaintexts)
    for name, k, n, a, p_, c in (
            ("main.c:276 RFC-8452 #1", "ee8e1ed9ff2540ae8f2ba9f50bc2f27c", "752abad3e0afb5f434dc4310", "6578616d706c65",
             "48656c6c6f20776f726c64", "5d349ead175ef6b1def6fd4fbcdeb7e4793f4a1d7e4faa70100af1"),
            ("main.c:286 RFC-8452 #2", "01000000000000000000000000000000", "030000000000000000000000", "01",
             "0200000000000000000000000000000003000000000000000000000000000000",
             "620048ef3c1e73e57e02bb8562c416a319e73e4caac8e96a1ecb2933145a1d71e6af6a7f87287da059a71684ed3498e1")):
        kk, nn, aa, pp = (bytes.fromhex(x) for x in (k, n, a, p_))
        assert r128.gcmsiv_encrypt(kk, nn, aa, pp).hex() == c
        kats.append(dict(name=name, mode="gcmsiv", keybits=128, key=k, pt=p_, nonce=n, aad=a, expect=c))
    assert r128.cmac(key[:16], pt).hex() == MAIN_KATS["cmac128"]
    add("main.c:181 CMAC", "cmac", 128, key[:16], {}, MAIN_KATS["cmac128"])
    assert r128.ccm_encrypt(key[:16], iv[:11], aad, pt).hex() == MAIN_KATS["ccm128"]
    add("main.c:198 CCM", "ccm", 128, key[:16], {"nonce": iv[:11].hex(), "aad": aad.hex()}, MAIN_KATS["ccm128"])
    return kats


LENGTHS = [0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 57, 64, 255, 256, 257, 1000,
           4095, 4096, 4097, 4111, 16384, 65541]


def ref_vectors(orc):
    rng = random.Random(0x75414553)
    vecs = []
    for bits in (128, 192, 256):
        ref = Reference(bits)
        kb = bits // 8
        for n in LENGTHS:
            seed = rng.getrandbits(32)
            data = splitmix(orc, seed, n)
            key = rng.randbytes(kb)
            base = dict(keybits=bits, len=n, seed=seed)
            # ECB (N1: zero padded output on encrypt; 0x1d on ragged decrypt)
            ct = ref.ecb_encrypt(key, data)
            rc, back = ref.ecb_decrypt(key, ct[:n] if n % 16 else ct)
            vecs.append(dict(base, mode="ecb", key=key.hex(), out=enc_out(ct), dec_rc=rc))
            # CTR
            iv = rng.randbytes(12)
            vecs.append(dict(base, mode="ctr", key=key.hex(), iv=iv.hex(),
                             out=enc_out(ref.ctr_encrypt(key, iv, data))))
            # XTS (N5: len < 16 -> rc 1, output untouched)
            keys, tweak = rng.randbytes(2 * kb), rng.randbytes(16)
            rc, ct = ref.xts(keys, tweak, data, True)
            v = dict(base, mode="xts", key=keys.hex(), tweak=tweak.hex(), rc=rc)
            if rc == 0:
                v["out"] = enc_out(ct)
                rc2, pt2 = ref.xts(keys, tweak, ct, False)
                assert rc2 == 0 and pt2 == data
            vecs.append(v)
            # GCM (slow bit-serial GHASH in the reference: cap the size)
            if n <= 16384:
                nonce, aad = rng.randbytes(12), rng.randbytes(rng.choice([0, 1, 13, 16, 20, 32, 90]))
                ct = ref.gcm_encrypt(key, nonce, aad, data)
                rc, pt2 = ref.gcm_decrypt(key, nonce, aad, ct)
                assert rc == 0 and pt2 == data
                bad = bytearray(ct); bad[-1] ^= 1
                rcb, ptb = ref.gcm_decrypt(key, nonce, aad, bytes(bad))
                assert rcb == 0x1A and ptb == b"\xcc" * n      # N7
                vecs.append(dict(base, mode="gcm", key=key.hex(), nonce=nonce.hex(),
                                 aad=aad.hex(), out=enc_out(ct)))
            # CBC (CS3), CFB, OFB
            iv16 = rng.randbytes(16)
            rc, ct = ref.cbc(key, iv16, data, True)
            v = dict(base, mode="cbc", key=key.hex(), iv=iv16.hex(), rc=rc)
            if rc == 0:
                v["out"] = enc_out(ct)
                assert ref.cbc(key, iv16, ct, False) == (0, data)
            vecs.append(v)
            ct = ref.cfb(key, iv16, data, True)
            assert ref.cfb(key, iv16, ct, False) == data
            vecs.append(dict(base, mode="cfb", key=key.hex(), iv=iv16.hex(), out=enc_out(ct)))
            vecs.append(dict(base, mode="ofb", key=key.hex(), iv=iv16.hex(), out=enc_out(ref.ofb(key, iv16, data))))
            # OCB
            nonce, aad = rng.randbytes(12), rng.randbytes(rng.choice([0, 1, 15, 16, 17, 100, 4096 + 5]))
            ct = ref.ocb_encrypt(key, nonce, aad, data)
            assert ref.ocb_decrypt(key, nonce, aad, ct) == (0, data)
            vecs.append(dict(base, mode="ocb", key=key.hex(), nonce=nonce.hex(), aad=aad.hex(), out=enc_out(ct)))
            # GCM-SIV (bit-serial POLYVAL in the reference: cap the size)
            if n <= 16384:
                nonce, aad = rng.randbytes(12), rng.randbytes(rng.choice([0, 1, 15, 16, 17, 100]))
                ct = ref.gcmsiv_encrypt(key, nonce, aad, data)
                assert ref.gcmsiv_decrypt(key, nonce, aad, ct) == (0, data)
                vecs.append(dict(base, mode="gcmsiv", key=key.hex(), nonce=nonce.hex(), aad=aad.hex(),
                                 out=enc_out(ct)))
            # CMAC and CCM
            if n <= 16384:
                vecs.append(dict(base, mode="cmac", key=key.hex(), out=enc_out(ref.cmac(key, data))))
                nonce11 = rng.randbytes(11)
                aad = rng.randbytes(rng.choice([0, 1, 13, 14, 15, 16, 30, 64, 300]))
                ct = ref.ccm_encrypt(key, nonce11, aad, data)
                rc, back = ref.ccm_decrypt(key, nonce11, aad, ct)
                assert rc == 0 and back == data
                vecs.append(dict(base, mode="ccm", key=key.hex(), nonce=nonce11.hex(), aad=aad.hex(),
                                 out=enc_out(ct)))
    # counter carry (N2): start the 56-bit counter near its wrap via the PT-side
    # trick is impossible through the 12-byte-IV API (counter always starts at
    # 1), so the carry is pinned by the 65541-byte CTR vectors (4097 blocks:
    # byte 15 wraps 16 times, byte 14 increments) and by test_oracle's
    # comparison of ctr56 arithmetic with python integers.
    return vecs


def digests(orc, big):
    r128, r256 = Reference(128), Reference(256)
    key16 = bytes(range(16))
    key64 = bytes(range(64))
    nonce = bytes(range(0xF0, 0xFC))
    sha = lambda b: hashlib.sha256(b).hexdigest()
    d = {}
    # C1: AES-128-ECB 4 KiB seed 1
    d["C1_ecb128_4KiB"] = dict(sha256=sha(r128.ecb_encrypt(key16, splitmix(orc, 1, 4096))),
                               survey="35065e081a72459058663f1b9ffc2ae298a808909b970567a02c52653fa72924")
    ct = r128.ctr_encrypt(key16, nonce, splitmix(orc, 2, 1 << 20))
    d["ctr128_1MiB_seed2"] = dict(sha256=sha(ct), head=ct[:32].hex(),
                                  survey="d0de5a73639234af5a0eb9a5a85c77719e4523c865555a4b55e68f3f61f8c34f")
    # XTS-256, 4 KiB sectors, seed 3 continuous, tweak = LE64(sector) || 0^8
    pt = splitmix(orc, 3, 3 * 4096)
    secs = b"".join(r256.xts(key64, s.to_bytes(16, "little"), pt[s * 4096:(s + 1) * 4096])[1]
                    for s in range(3))
    d["xts256_sector0"] = dict(sha256=sha(secs[:4096]), head=secs[:16].hex(),
                               survey="14d91b1f18d016a3d3c5eda716a25f5db3b551493b53156b8d958e4aa4c24eb4")
    d["xts256_sectors0_2"] = dict(sha256=sha(secs),
                                  survey="8c9780eb74e3a99e8806893a086df0554cce74892d8eff287f1210e9807a579c")
    d["xts256_sector2"] = dict(sha256=sha(secs[8192:]),
                               survey="4b896eaa6df5f5f8bef1e0df80ccd67bcc934b3a1e264ce069f990af9827d50b")
    ct = r128.gcm_encrypt(key16, nonce, b"", splitmix(orc, 4, 1 << 20))
    d["gcm128_1MiB_seed4"] = dict(tag=ct[-16:].hex(), sha256_ct_tag=sha(ct), sha256_ct=sha(ct[:-16]),
                                  survey_tag="2d85782c7ee81e5a4d3465f7ba647310",
                                  survey="51bdf5b9cdd300f1764429d082a32f53e76b0963cb0cd5949e8d970891094e3f")
    # full-size BASELINE configs: values measured with the compiled reference
    # during the survey; recomputed only with --big
    d["C2_ctr128_1GiB_seed2"] = dict(
        sha256="c0504cbc0d799823405099c425509d4b86b2208135a6b8ad424ef02a31398d40",
        tail="67b2827a71b064885cf195bddc5f72542b78e93c82a83498405197fa73e73219",
        source="SURVEY.md 8d (compiled reference)")
    d["C3_xts256_2p20_sectors_seed3"] = dict(
        sha256="ae8ad1b9ff34a249ada238324915b6f3a96ba7776ca5b5387b5afe2f110d0be2",
        source="SURVEY.md 8d (compiled reference)")
    d["C4_gcm128_1GiB_seed4"] = dict(
        tag="84e8fd11a2f269c6800b2e2924c40635",
        sha256_ct_tag="57d46720f63a632e37ae45207cf7471548a2fb912c4753b7f8909f0c992c9cd9",
        source="SURVEY.md 8d (compiled reference)")
    d["C5_ctr128_8GiB_seed2"] = dict(
        sha256="5066df5498cacd8b639f3e4fcaed7f6a64c28460702890d4939ab49e66717c46",
        source="SURVEY.md 8d (compiled reference)")
    # C5 per shard: GPU g of BASELINE configs[4] owns bytes [g GiB, (g+1) GiB) of the 8 GiB stream and starts at
    # counter offset g * 2^26.  Computed with the REFERENCE built with PRESET_COUNTER 1
    # (micro_aes.h:100, micro_aes.c:965-966: the caller passes the whole counter block), see c5_shards();
    # kept from the last --big run otherwise
    try:
        with open(os.path.join(HERE, "digests.json")) as f:
            old = json.load(f)
    except Exception:
        old = {}
    for g in range(8):
        if "C5_shard_%d" % g in old:
            d["C5_shard_%d" % g] = old["C5_shard_%d" % g]
    for k, v in d.items():
        if "survey" in v:
            assert v["sha256" if "sha256" in v else "sha256_ct_tag"] == v["survey"], k
    if big:
        h = hashlib.sha256()
        step = 1 << 26
        tail = b""
        # 1 GiB in 64 MiB slices: CTR slices are independent given the block
        # offset, and the reference API always starts at counter 1, so feed it
        # the whole stream in one call instead (needs ~2 GiB RAM).
        pt = splitmix(orc, 2, 1 << 30)
        ct = r128.ctr_encrypt(key16, nonce, pt)
        assert sha(ct) == d["C2_ctr128_1GiB_seed2"]["sha256"]
        assert ct[-32:].hex() == d["C2_ctr128_1GiB_seed2"]["tail"]
        d["C2_ctr128_1GiB_seed2"]["recomputed"] = True
        del h, step, tail, pt, ct
        d.update(c5_shards(orc, d))
    return d


def c5_shards(orc, d):
    """SHA-256 of each of the eight 1 GiB shards of the C5 stream, by the reference's own AES_CTR_encrypt
    (PRESET_COUNTER build, counter block = iv || BE32(1 + g * 2^26)), eight forked processes; the
    concatenation is checked against the survey's digest of the whole 8 GiB stream."""
    import numpy as np
    import ctypes as C
    ref = Reference(128, preset_counter=True)
    key16 = bytes(range(16))
    nonce = bytes(range(0xF0, 0xFC))
    n = 1 << 30
    tmp = "/tmp/uaes_c5_shards"
    os.makedirs(tmp, exist_ok=True)
    pids = []
    for g in range(8):
        pid = os.fork()
        if pid == 0:
            buf = np.empty(n, dtype=np.uint8)
            orc.splitmix_into(2, buf, word0=g * (n // 8))
            out = np.empty(n, dtype=np.uint8)
            ctr = nonce + (1 + g * (n // 16)).to_bytes(4, "big")
            ref.L.AES_CTR_encrypt(key16, ctr, C.c_void_p(buf.ctypes.data), n, C.c_void_p(out.ctypes.data))
            out.tofile(os.path.join(tmp, "shard%d.bin" % g))
            os._exit(0)
        pids.append(pid)
    for pid in pids:
        assert os.waitpid(pid, 0)[1] == 0
    res, whole = {}, hashlib.sha256()
    for g in range(8):
        h = hashlib.sha256()
        with open(os.path.join(tmp, "shard%d.bin" % g), "rb") as f:
            while True:
                b = f.read(1 << 26)
                if not b:
                    break
                h.update(b)
                whole.update(b)
        os.remove(os.path.join(tmp, "shard%d.bin" % g))
        res["C5_shard_%d" % g] = dict(sha256=h.hexdigest(), counter_offset=g * (n // 16),
                                      source="compiled reference, PRESET_COUNTER 1 (make_fixtures.py --big)")
    assert whole.hexdigest() == d["C5_ctr128_8GiB_seed2"]["sha256"], "shards do not concatenate to the C5 stream"
    assert res["C5_shard_0"]["sha256"] == d["C2_ctr128_1GiB_seed2"]["sha256"]
    return res


def ecb_padding_vectors(orc):
    """AES_ECB_encrypt of reference builds with AES_PADDING 1 (PKCS#7, the AES-192 build that
    main.c:139 tests) and 2 (ISO/IEC 7816-4), padBlock micro_aes.c:610-621, plus the main.c KAT"""
    rng = random.Random(0x70616421)
    vecs = []
    main_c_192 = ("af1893f0fbb09a437f6b0fd4f49778907bb85cccf1e9d2e3ebe5bae935107868"
                  "c6d72cb2ca375c12ce6b6b1141141fd0d268d14db351d6805aabb99427341da9")
    key = bytes.fromhex("279fb74a7572135e8f9b8ef6d1eee00369c4e0d86a7b0430d8cdb78070b4c55a")
    pt = bytes.fromhex("c9f775baafa36c25cd610d3c75a482eadda97ca4864cdfe06eaf70a0ec0d7191"
                       "d55027cf8f900214e634412583ff0b478EA2B7CA516745BFEA")
    r = Reference(192, padding=1)
    assert r.ecb_encrypt(key[:24], pt).hex() == main_c_192           # main.c:86-87,139-141
    vecs.append(dict(name="main.c:139 ECB AES-192 PKCS#7", keybits=192, padding=1, key=key[:24].hex(),
                     pt=pt.hex(), out=main_c_192))
    for bits, padding in ((192, 1), (128, 2)):
        ref = Reference(bits, padding=padding)
        for n in [0, 1, 15, 16, 17, 31, 32, 33, 255, 256, 4095, 4096, 4097, 65541]:
            seed = rng.getrandbits(32)
            k = rng.randbytes(bits // 8)
            vecs.append(dict(keybits=bits, padding=padding, len=n, seed=seed, key=k.hex(),
                             out=enc_out(ref.ecb_encrypt(k, splitmix(orc, seed, n)))))
    return vecs


def build_variant_vectors(orc):
    """AES_CBC_* of reference builds with CTS 0 and each AES_PADDING (micro_aes.h:56,79; micro_aes.c:704-733,
    :753-761) and AES_CTR_encrypt of builds with other CTR_IV_LENGTH / CTR_START_VALUE (micro_aes.h:98-99), on
    seeded inputs, plus main.c's own CTS 0 known answer (main.c:36-40, :149)"""
    rng = random.Random(0x6e6f6374)
    cbc, ctr = [], []
    ref = Reference.of_variant("nocts")
    key, iv, pt = bytes.fromhex(MAIN_KEY)[:16], bytes.fromhex(MAIN_IV), bytes.fromhex(MAIN_PT)
    main_c = ("65c48fdf9fbd626128f2d8bac3f7125175e7f4821fda026370011632779d7403"
              "7E9E2D298E154BC42Dc7a9bc419b915dc119ef461ac4e1bc8a7e36bf92b3b3d1").lower()
    rc, out = ref.cbc_nocts(key, iv, pt, True)
    assert rc == 0 and out.hex() == main_c                            # main.c:30-40,147-149 with CTS 0
    cbc.append(dict(name="main.c:149 CBC AES-128 CTS 0", variant="nocts", keybits=128, padding=0, key=key.hex(),
                    iv=iv.hex(), pt=pt.hex(), out={"hex": main_c}))
    for v, (bits, padding) in Reference.NOCTS.items():
        ref = Reference.of_variant(v)
        for n in [0, 1, 15, 16, 17, 31, 32, 33, 48, 255, 256, 4095, 4096, 4097, 65541]:
            seed = rng.getrandbits(32)
            k, ivv = rng.randbytes(bits // 8), rng.randbytes(16)
            data = splitmix(orc, seed, n)
            rc, out = ref.cbc_nocts(k, ivv, data, True)
            assert rc == 0
            rcd, back = ref.cbc_nocts(k, ivv, out, False)
            assert rcd == 0 and back[:n] == data
            cbc.append(dict(variant=v, keybits=bits, padding=padding, len=n, seed=seed, key=k.hex(), iv=ivv.hex(),
                            out=enc_out(out), ragged_decrypt_rc=ref.cbc_nocts(k, ivv, data, False)[0] if n % 16 else 0))
    for v, (bits, ivl, start) in Reference.CTRV.items():
        ref = Reference.of_variant(v)
        for n in [0, 1, 15, 16, 17, 33, 255, 4096, 4097, 65541]:
            seed = rng.getrandbits(32)
            k, ivv = rng.randbytes(bits // 8), rng.randbytes(ivl)
            if n == 4097:
                ivv = ivv[:9] + b"\xff" * (ivl - 9) if ivl > 9 else ivv      # counter bytes about to carry
            ctr.append(dict(variant=v, keybits=bits, iv_length=ivl, start_value=start, len=n, seed=seed, key=k.hex(),
                            iv=ivv.hex(), out=enc_out(ref.ctr_encrypt(k, ivv, splitmix(orc, seed, n)))))
    return dict(cbc_nocts=cbc, ctr_iv=ctr)


def main():
    if "--variants" in sys.argv:           # only the build-variant vectors (leaves the other fixtures as they are)
        build()
        with open(os.path.join(HERE, "build_variant_vectors.json"), "w") as f:
            json.dump(build_variant_vectors(Oracle()), f, indent=0)
        print("build_variant_vectors.json written")
        return
    big = "--big" in sys.argv
    build()
    orc = Oracle()
    for bits in (128, 192, 256):
        filter_gcm(os.path.join(REF_TV, "GcmEncryptExtIV%d.rsp" % bits),
                   os.path.join(HERE, "GcmEncryptExtIV%d.rsp" % bits))
    plain = ["XTSGenAES128.rsp", "XTSGenAES256.rsp"]
    plain += ["CMACGenAES%d.rsp" % b for b in (128, 192, 256)] + ["VNT%d.rsp" % b for b in (128, 192, 256)]
    plain += ["SIV_GCM_ACVP.tv", "OCB_AES128.tv"]
    for name in plain:
        with open(os.path.join(REF_TV, name)) as f, open(os.path.join(HERE, name), "w") as g:
            g.write(f.read())
    with open(os.path.join(HERE, "main_kats.json"), "w") as f:
        json.dump(main_kats(), f, indent=1)
    with open(os.path.join(HERE, "ref_vectors.json"), "w") as f:
        json.dump(ref_vectors(orc), f, indent=0)
    with open(os.path.join(HERE, "ecb_padding_vectors.json"), "w") as f:
        json.dump(ecb_padding_vectors(orc), f, indent=0)
    with open(os.path.join(HERE, "build_variant_vectors.json"), "w") as f:
        json.dump(build_variant_vectors(orc), f, indent=0)
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump(digests(orc, big), f, indent=1)
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
