"""CPU tests of the product's host side: the C-ABI libraries load and export
every symbol include/*.h declares, the host key schedule is right, the error
behaviour without a GPU is loud, and nothing in the product references the
oracle."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

import micro_aes_amd as uaes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"UAES_STATIC_INLINE[^{;]*\{.*?\n\}", "", text, flags=re.S)      # header-only wrappers are not exports
    return sorted(set(re.findall(r"\b((?:uaes|AES|GCM_SIV)_[A-Za-z0-9_]+)\s*\(", text)))


def test_runtime_library_exports_every_declared_symbol():
    names = declared_functions("uaes_hip.h")
    assert len(names) >= 20 and set(uaes.EXPORTS) == set(names)
    L = C.CDLL(uaes.lib_path())
    for n in names:
        assert getattr(L, n) is not None


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_compat_libraries_export_reference_api(bits):
    names = declared_functions("micro_aes.h")
    assert set(names) == set(uaes.COMPAT_EXPORTS)
    L = C.CDLL(uaes.lib_path("libmicro_aes_hip_%d.so" % bits))
    for n in names:
        assert getattr(L, n) is not None


def test_compat_header_compiles_as_c89_and_matches_reference_constants(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "micro_aes.h"\n#include <stdio.h>\n'
                   'int main(void){printf("%d %d %d %d %d %d %d %d %d\\n", AES_KEYLENGTH, GCM_NONCE_LEN, GCM_TAG_LEN,'
                   'CTR_IV_LENGTH, CTR_START_VALUE, (int)M_DATALENGTH_ERROR, M_AUTHENTICATION_ERROR,'
                   'M_DECRYPTION_ERROR, M_ENCRYPTION_ERROR);return ECB&&CTR&&XTS&&GCM&&CMAC&&CCM&&CBC&&CTS&&CFB&&OFB&&GCM_SIV&&OCB&&!EAX&&!KWA?0:1;}\n')
    for bits, kl in ((128, 16), (192, 24), (256, 32)):
        exe = tmp_path / ("t%d" % bits)
        subprocess.run(["gcc", "-std=c89", "-pedantic", "-Wall", "-Werror", "-DAES___=%d" % bits,
                        "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
        out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
        assert [int(x) for x in out] == [kl, 12, 16, 12, 1, 1, 0x1A, 0x1D, 0x1E]


def test_compat_header_binds_function_pointers_under_the_callers_switches(tmp_path):
    """-DGCM_NONCE_LEN=n / -DPRESET_COUNTER=1 / -DAES_PADDING=1 and the other length constants (-DGCM_TAG_LEN,
    -DCCM_NONCE_LEN, -DCCM_TAG_LEN, -DOCB_NONCE_LEN, -DOCB_TAG_LEN) are compile-time switches of the CALLER's
    build (micro_aes.h:79,100,103-116).  The header must bind a function POINTER -- not only a direct call --
    to the matching entry point: object-like macros, never function-like ones; and stay C89-clean."""
    src = tmp_path / "p.c"
    src.write_text('#include "micro_aes.h"\n'
                   'typedef void (*gcm_fn)(const uint8_t*, const uint8_t*, const void*, const size_t, const void*, const size_t, void*);\n'
                   'typedef void (*ctr_fn)(const uint8_t*, const uint8_t*, const void*, const size_t, void*);\n'
                   'typedef void (*ecb_fn)(const uint8_t*, const void*, const size_t, void*);\n'
                   'gcm_fn table_g[3] = { AES_GCM_encrypt, AES_CCM_encrypt, AES_OCB_encrypt };\n'
                   'ctr_fn table_c[2] = { AES_CTR_encrypt, AES_CTR_decrypt };\n'
                   'ecb_fn table_e[1] = { AES_ECB_encrypt };\n'
                   'int main(void) { return (table_g[0] && table_c[0] && table_e[0] && GCM_NONCE_LEN == 7 && GCM_TAG_LEN == 16\n'
                   '                         && CCM_NONCE_LEN == 13 && CCM_TAG_LEN == 8 && OCB_NONCE_LEN == 12 && OCB_TAG_LEN == 12) ? 0 : 1; }\n')
    obj = tmp_path / "p.o"
    subprocess.run(["gcc", "-std=c89", "-pedantic", "-Wall", "-Werror", "-DGCM_NONCE_LEN=7", "-DPRESET_COUNTER=1",
                    "-DAES_PADDING=1", "-DCCM_NONCE_LEN=13", "-DCCM_TAG_LEN=8", "-DOCB_TAG_LEN=12", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(obj)], check=True)
    syms = subprocess.run(["nm", "-u", str(obj)], check=True, capture_output=True, text=True).stdout
    assert "AES_GCM_encrypt_lens" in syms and "AES_CTR_encrypt_preset" in syms and "AES_CTR_decrypt_preset" in syms
    assert "AES_ECB_encrypt_pkcs7" in syms and "AES_CCM_encrypt_lens" in syms and "AES_OCB_encrypt_lens" in syms
    assert not re.search(r"\bAES_GCM_encrypt$|\bAES_CTR_encrypt$|\bAES_ECB_encrypt$|\bAES_CCM_encrypt$|\bAES_OCB_encrypt$",
                         syms, flags=re.M)
    # CTS 0 / other CTR constants bind likewise (micro_aes.h:56, :98-99)
    src2 = tmp_path / "q.c"
    src2.write_text('#include "micro_aes.h"\n'
                    'typedef void (*ctr_fn)(const uint8_t*, const uint8_t*, const void*, const size_t, void*);\n'
                    'typedef char (*cbc_fn)(const uint8_t*, const uint8_t*, const void*, const size_t, void*);\n'
                    'ctr_fn table_c[2] = { AES_CTR_encrypt, AES_CTR_decrypt };\n'
                    'cbc_fn table_b[2] = { AES_CBC_encrypt, AES_CBC_decrypt };\n'
                    'int main(void) { return (table_c[0] && table_b[0] && CTR_IV_LENGTH == 8 && CTR_START_VALUE == 7 && !CTS) ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c89", "-pedantic", "-Wall", "-Werror", "-DCTS=0", "-DAES_PADDING=2", "-DCTR_IV_LENGTH=8",
                    "-DCTR_START_VALUE=7", "-I", os.path.join(ROOT, "include"), "-c", str(src2), "-o", str(obj)], check=True)
    syms2 = subprocess.run(["nm", "-u", str(obj)], check=True, capture_output=True, text=True).stdout
    assert "AES_CBC_encrypt_nocts_iso7816" in syms2 and "AES_CBC_decrypt_nocts" in syms2 and "AES_CTR_encrypt_iv" in syms2
    assert not re.search(r"\bAES_CBC_encrypt$|\bAES_CBC_decrypt$|\bAES_CTR_encrypt$", syms2, flags=re.M)
    bad = subprocess.run(["gcc", "-std=c89", "-DCTR_IV_LENGTH=17", "-I", os.path.join(ROOT, "include"), "-c", str(src2), "-o", str(obj)],
                         capture_output=True, text=True)
    assert bad.returncode != 0 and "uaes_ctr_lengths_ok" in bad.stderr
    # a length outside the mode's range does not compile
    bad = subprocess.run(["gcc", "-std=c89", "-DCCM_TAG_LEN=5", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(obj)],
                         capture_output=True, text=True)
    assert bad.returncode != 0 and "uaes_ccm_lengths_ok" in bad.stderr


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_host_key_schedule(orc, bits):
    """uaes_expand_key == the oracle's schedule (itself pinned to the reference);
    the decrypt schedule is the equivalent-inverse form of the same keys."""
    import random
    rnd = random.Random(bits)
    L = uaes.engine()
    for _ in range(10):
        key = rnd.randbytes(bits // 8)
        ek, dk = (C.c_uint32 * 60)(), (C.c_uint32 * 60)()
        nr = L.uaes_expand_key(bits, key, ek, dk)
        assert nr == bits // 32 + 6
        ks = (C.c_uint8 * 244)()
        assert orc.L.orc_setkey(ks, key, bits) == 0
        want = bytes(ks)[4: 4 + 16 * (nr + 1)]
        assert bytes(ek)[: 16 * (nr + 1)] == want
        d = bytes(dk)
        assert d[:16] == want[16 * nr:] and d[16 * nr: 16 * nr + 16] == want[:16]
        # middle keys: MixColumns(dk[i]) == ek[nr-i]  <=>  dk[i] = InvMixColumns(ek[nr-i])
        def mixcol(col):
            x2 = lambda a: ((a << 1) ^ (0x1b if a & 0x80 else 0)) & 0xff
            a = list(col)
            t = a[0] ^ a[1] ^ a[2] ^ a[3]
            return bytes(a[i] ^ t ^ x2(a[i] ^ a[(i + 1) % 4]) for i in range(4))
        for i in range(1, nr):
            for c in range(4):
                col = d[16 * i + 4 * c: 16 * i + 4 * c + 4]
                assert mixcol(col) == want[16 * (nr - i) + 4 * c: 16 * (nr - i) + 4 * c + 4]
    assert L.uaes_expand_key(100, bytes(16), None, None) == -2
    assert b"keybits" in L.uaes_last_error()


def test_fails_loudly_without_gpu():
    """no silent CPU path: on a box without a HIP device every data call errors"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(uaes.EngineError, match="no usable HIP device"):
        uaes.AES_CTR_encrypt(bytes(16), bytes(12), b"abc")
    with pytest.raises(uaes.EngineError):
        uaes.selftest()


def test_void_functions_report_engine_failures_to_the_installed_handler():
    """the reference's `void` API cannot return an error; the compat library hands an engine
    failure to a replaceable handler (default: print + abort).  Without a GPU every call fails,
    which is exactly the case to observe here."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = C.CDLL(uaes.lib_path("libmicro_aes_hip_128.so"))
    HANDLER = C.CFUNCTYPE(None, C.c_char_p, C.c_int, C.c_char_p)
    seen = []
    cb = HANDLER(lambda fn, rc, msg: seen.append((fn.decode(), rc, msg.decode())))
    L.uaes_compat_set_failure_handler.restype = C.c_void_p
    L.uaes_compat_set_failure_handler.argtypes = [HANDLER]
    assert L.uaes_compat_set_failure_handler(cb) is None          # the default was installed
    out = (C.c_uint8 * 32)()
    L.AES_CTR_encrypt(bytes(16), bytes(12), b"0123456789abcdef", C.c_size_t(16), out)
    L.AES_ECB_encrypt_pkcs7(bytes(16), b"0123456789abcdef", C.c_size_t(16), out)
    assert [s[0] for s in seen] == ["AES_CTR_encrypt", "AES_ECB_encrypt"]
    assert all(rc == -1 and "no usable HIP device" in msg for _, rc, msg in seen)
    # char-returning functions keep returning the reference's codes
    L.AES_XTS_encrypt.restype = C.c_char
    assert ord(L.AES_XTS_encrypt(bytes(32), bytes(16), b"0123456789abcdef", C.c_size_t(16), out)) == 0x1E
    # the default handler aborts the process
    code = ("import ctypes as C; L = C.CDLL(%r); o = (C.c_uint8 * 32)(); "
            "L.AES_CTR_encrypt(bytes(16), bytes(12), b'0123456789abcdef', C.c_size_t(16), o)"
            % uaes.lib_path("libmicro_aes_hip_128.so"))
    r = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == -6 and "AES_CTR_encrypt failed" in r.stderr       # SIGABRT


def test_python_wrappers_reject_wrong_sized_nonces():
    """the C side copies a fixed number of bytes: short or long iv / nonce / tweak / counter
    blocks are refused before any call is made (no GPU needed)"""
    k = bytes(16)
    for call in (lambda: uaes.AES_CTR_encrypt(k, bytes(11), b"x"), lambda: uaes.AES_CTR_encrypt(k, bytes(16), b"x"),
                 lambda: uaes.ctr_xcrypt_at(k, bytes(12), 0, b"x"), lambda: uaes.AES_XTS_encrypt(bytes(32), bytes(8), bytes(16)),
                 lambda: uaes.AES_GCM_encrypt(k, b"", b"", b"x"), lambda: uaes.AES_GCM_decrypt(k, b"", b"", bytes(17)),
                 lambda: uaes.gcm_encrypt_dev(k, bytes(16), None, None, 0, None),
                 lambda: uaes.AES_CCM_encrypt(k, bytes(6), b"", b"x"), lambda: uaes.AES_CCM_encrypt(k, bytes(14), b"", b"x"),
                 lambda: uaes.AES_CCM_encrypt(k, bytes(11), b"", b"x", tag_len=5), lambda: uaes.AES_CCM_decrypt(k, bytes(11), b"", bytes(9), tag_len=2),
                 lambda: uaes.AES_OCB_encrypt(k, bytes(16), b"", b"x"), lambda: uaes.AES_OCB_encrypt(k, b"", b"", b"x"),
                 lambda: uaes.AES_OCB_encrypt(k, bytes(12), b"", b"x", tag_len=17), lambda: uaes.AES_GCM_encrypt(k, bytes(12), b"", b"x", tag_len=0),
                 lambda: uaes.GCM_SIV_encrypt(k, bytes(13), b"", b"x"), lambda: uaes.AES_CBC_encrypt(k, bytes(12), bytes(16)),
                 lambda: uaes.GcmStream(k, bytes(16)), lambda: uaes.ghash(bytes(15), b"", b"x"),
                 lambda: uaes.AES_ECB_encrypt(k, b"x", padding=3)):
        with pytest.raises(ValueError):
            call()


def test_product_does_not_touch_the_oracle():
    """the shipped path must not import, link or open anything under oracle/"""
    pkg = os.path.join(ROOT, "micro-aes_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle/" not in text.replace("the oracle", "") or "oracle/" not in re.sub(r"(#|//|/\*|\*|\"\"\").*", "", text), f
                assert "pyoracle" not in text and "liboracle" not in text and "uaes_oracle" not in text, f
    out = subprocess.run(["ldd", uaes.lib_path()], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_bulk_kernels_have_no_private_segment_and_only_gfx950_code():
    """every grid-wide kernel of the library must run out of registers and LDS alone: a private
    (scratch) segment -- e.g. from a register array indexed by a run-time value in a one-thread tail
    case -- makes every wave of the launch set up scratch.  Read from the code objects' metadata
    (tools/kernel_resources.py); the same pass checks that only gfx950 code objects are shipped."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    ks = kernel_resources.kernels()
    assert len(ks) > 100
    bulk = re.compile(r"\bk_(ecb|ctr|ctr_shared2|xts|xts_tweaks|xts_expand|xts_cts|fb_dec|chain_batch|gcm_fused|"
                      r"ghash_pass|gcm_chunks|ocb|wipe_if_failed)\b")
    seen = set()
    for k in ks:
        m = bulk.search(k["name"])
        if not m:
            continue
        seen.add(m.group(1))
        assert k["scratch"] == 0 and k["vgpr_spill"] == 0, (k["name"][:80], k["scratch"], k["vgpr_spill"])
        assert k["vgpr"] <= 128, (k["name"][:80], k["vgpr"])          # 4 waves per SIMD at least
    assert {"ecb", "ctr", "ctr_shared2", "xts", "fb_dec", "chain_batch", "gcm_fused", "ghash_pass", "gcm_chunks", "ocb"} <= seen
    # the one-workgroup kernels of the short calls as well: a scratch access is a round trip to memory in the middle
    # of a latency-bound chain (round 3: k_gcm_small, k_siv_small, k_cmac, k_ccm_tag, k_ocb_small / _final had
    # step arrays and byte buffers indexed at run time there).  The one exception is the decrypting record kernel,
    # which keeps four registers of held plaintext across its hash.
    for k in ks:
        if re.search(r"\bk_gcm_records<\d+, true>", k["name"]):
            assert k["scratch"] <= 32 and k["vgpr_spill"] <= 8, (k["name"][:80], k["scratch"], k["vgpr_spill"])
        else:
            assert k["scratch"] == 0 and k["vgpr_spill"] == 0, (k["name"][:80], k["scratch"], k["vgpr_spill"])


def test_ctr_hot_loop_instruction_budget():
    """DESIGN.md section 6 prices AES-128-CTR in instructions per block: 128 table lookups and about 210 VALU operations.
    Re-derived here from the SHIPPED code object (llvm-objdump of libuaes_hip.so, tools/kernel_resources.py --disasm),
    so that a compiler upgrade or an innocent edit cannot add instructions to the loop unnoticed.  One trip of the hot
    loop of k_ctr_shared2<10> = the body twice = 4 blocks per lane: 2 x 2 x 8 rounds x 16 lookups and nothing else --
    since round 5 the loop runs chunk by chunk (eight iterations = one fill of the U-buffer), so wave 0's refill sits in
    the outer loop, and the lane constants L0..L3 are made once per launch (the host cuts a text where counter bits 40..47
    move), so the loop carries no conditionally defined value: no v_mov at all."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    dis = kernel_resources.disassemble()
    names = [n for n in dis if n.startswith("void k_ctr_shared2<10>")]
    assert len(names) == 1
    text, counts = kernel_resources.loop_report(names[0], dis[names[0]])
    assert counts["ds_read_b32"] == 2 * 2 * 8 * 16, counts["ds_read_b32"]
    valu = sum(n for op, n in counts.items() if op.startswith("v_"))
    # per block: 7 rounds x 24 + the last round's 16 + 12 + 4 (round-3 entry) + 4 (text xor) = 204, + the loop's bookkeeping
    assert valu <= 4 * 204 + 20, valu
    assert counts.get("buffer_load_dwordx4", 0) + counts.get("global_load_dwordx4", 0) == 4
    assert counts.get("v_mov_b32_e32", 0) == 0            # nothing loop-carried is copied: the text buffers are renamed
    assert counts.get("s_barrier", 0) == 0                # (the refill and its barrier belong to the outer loop)
    # sixteen lookups per round and block: between two s_setprio 1 there must be exactly 16 ds_read_b32
    ops = [l.split()[1] for l in text.splitlines() if l.startswith("  ")]
    args = [l.split()[2] if len(l.split()) > 2 else "" for l in text.splitlines() if l.startswith("  ")]
    bursts, cur, on = [], 0, False
    for op, a in zip(ops, args):
        if op == "s_setprio":
            if on and a == "0":
                bursts.append(cur)
            on, cur = a == "1", 0
        elif op == "ds_read_b32" and on:
            cur += 1
    assert bursts and set(bursts) == {16}, sorted(set(bursts))


def test_the_table_of_arrangements_without_a_device():
    """csrc/uaes_plan.h as data (uaes_debug_plan): without a GPU the planners answer for a 256-CU MI355X.  Every
    arrangement has a name, the boundaries sit where DESIGN.md's table says, the switch (uaes_debug_plan_disable)
    makes the next arrangement take the call, and no size threshold is read from the environment any more."""
    L = uaes.engine()
    names = [L.uaes_debug_arrangement_name(k).decode() for k in range(18)]
    assert names == ["ecb.single", "ecb.tiled", "ctr.single", "ctr.quad", "ctr.striped", "xts.small", "xts.packed",
                     "xts.bulk", "gcm.small", "gcm.chunks", "gcm.twophase", "gcm.striped", "gcm.levels", "ocb.small", "ocb.runs",
                     "siv.small", "siv.chunks", "siv.levels"]
    assert L.uaes_debug_arrangement_name(18) == b"?"
    MIB = 1 << 20
    table = [("ecb", 4096, 0, 0, "ecb.single"), ("ecb", 1 << 30, 0, 0, "ecb.tiled"),
             ("ctr", 4096, 0, 0, "ctr.single"), ("ctr", 8 * MIB - 16, 0, 0, "ctr.single"), ("ctr", 9 * MIB, 0, 0, "ctr.striped"),
             ("ctr", 1 << 30, 0, 0, "ctr.striped"),
             ("xts", 4096, 1, 0, "xts.small"), ("xts", 8 * MIB, 1, 0, "xts.small"), ("xts", 8 * MIB + 16, 1, 0, "xts.bulk"),
             ("xts", 4096, 1024, 0, "xts.small"), ("xts", 4096, 1025, 0, "xts.bulk"), ("xts", 4096 + 16, 1025, 0, "xts.bulk"),
             ("xts", 4096, 1 << 20, 0, "xts.bulk"), ("xts", 512, 1 << 20, 0, "xts.packed"),
             ("gcm", 0, 0, 0, "gcm.small"), ("gcm", 2045 * 16, 0, 0, "gcm.small"), ("gcm", 2046 * 16, 0, 0, "gcm.chunks"),
             ("gcm", 16 * MIB, 0, 0, "gcm.chunks"), ("gcm", 16 * MIB + 16, 0, 0, "gcm.twophase"), ("gcm", 128 * MIB, 0, 0, "gcm.twophase"),
             ("gcm", 128 * MIB + 16, 0, 0, "gcm.striped"), ("gcm", 1 << 30, 0, 0, "gcm.striped"),
             ("gcm", 16 * MIB + 16, 0, 1, "gcm.chunks"), ("gcm", 512 * MIB, 0, 1, "gcm.chunks"), ("gcm", 512 * MIB + 16, 0, 1, "gcm.levels"),
             ("gcm", 1 << 30, 0, 2, "gcm.striped"), ("gcm", 1 << 20, 0, 3, "gcm.levels"),
             ("ocb", 16384, 0, 0, "ocb.small"), ("ocb", 16400, 0, 0, "ocb.runs"), ("ocb", 100, 70000, 0, "ocb.runs"),
             ("siv", 1000, 0, 0, "siv.small"), ("siv", 1 << 20, 0, 0, "siv.chunks"), ("siv", 600 * MIB, 0, 0, "siv.levels")]
    for mode, a, b, d, want in table:
        assert uaes.plan(mode, a, b, d)[0] == want, (mode, a, b, d, uaes.plan(mode, a, b, d))
    assert uaes.plan("gcm", 1 << 30)[1] == 3 and uaes.plan("gcm", 1 << 20)[1] == 1 and uaes.plan("gcm", 1 << 20, 0, 0, 4)[1] == 2
    try:
        L.uaes_debug_plan_disable(1 << uaes.arrangement_id("gcm.twophase"))
        assert uaes.plan("gcm", 64 * MIB)[0] == "gcm.striped"
        L.uaes_debug_plan_disable((1 << uaes.arrangement_id("gcm.twophase")) | (1 << uaes.arrangement_id("gcm.striped")))
        assert uaes.plan("gcm", 64 * MIB)[0] == "gcm.levels"
        L.uaes_debug_plan_disable(1 << uaes.arrangement_id("ctr.striped"))
        assert uaes.plan("ctr", 1 << 30)[0] == "ctr.quad"
        L.uaes_debug_plan_disable(0xffffffff)                 # the catch-all arrangements cannot be switched off
        assert [uaes.plan(m, 1 << 20)[0] for m in ("ecb", "ctr", "gcm", "ocb", "siv")] == ["ecb.tiled", "ctr.quad", "gcm.levels", "ocb.runs", "siv.levels"]
        assert uaes.plan("xts", 4096, 64)[0] == "xts.bulk"
    finally:
        L.uaes_debug_plan_disable(0)
    # the kernel layer reads no size threshold from the environment: the only getenv calls left are the documented hooks
    hooks = set()
    for f in ("uaes_kernels.hip", "uaes_gcm.hip", "uaes_gcm_records.hip", "uaes_siv.hip", "uaes_ghash.hip.h", "uaes_ocb.hip",
              "uaes_chain.hip", "uaes_mac.hip"):
        hooks |= set(re.findall(r'getenv\("(\w+)"\)', open(os.path.join(ROOT, "micro-aes_amd", "csrc", f)).read()))
    assert hooks == {"UAES_PLAN_DISABLE", "UAES_GCM_FOLD", "UAES_GCM_LOOK_TICKS"}, hooks
