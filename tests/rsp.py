"""Parsers for the NIST CAVP .rsp fixtures, applying the SAME case filters as
the reference's harness so that the case counts (375 / 800 / 600) are
themselves a parity check on the parser:

* GCM  -- testvectors/aes_testvectors_GCM.h:86: run a case only if
          Keylen == AES_KEYLENGTH, IVlen == GCM_NONCE_LEN (12 bytes) and
          Taglen >= GCM_TAG_LEN (16 bytes).
* XTS  -- testvectors/aes_testvectors_XTS.h:84: run a case only if the key is
          2*AES_KEYLENGTH bytes and DataUnitLen == 8 * len(PT) (whole bytes).
* CMAC -- testvectors/aes_testvectors_CMAC.h: every case whose key has
          AES_KEYLENGTH bytes; "Msg = 00" with Mlen = 0 is the empty message;
          only the first Tlen bytes of the MAC are compared.
* CCM  -- testvectors/aes_testvectors_CCM.h:83: key of AES_KEYLENGTH bytes,
          11-byte nonce (CCM_NONCE_LEN) and len(CT) - 16 == len(Payload),
          i.e. Tlen == 16 (CCM_TAG_LEN).
"""
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gcm_cases(keybits, iv_bytes=12):
    path = os.path.join(GOLDEN, "GcmEncryptExtIV%d.rsp" % keybits)
    hdr, cur, cases = {}, None, []
    with open(path) as f:
        for ln in f:
            ln = ln.strip()
            if ln.startswith("["):
                k, v = ln.strip("[]").split("=")
                hdr[k.strip()] = int(v)
            elif "=" in ln:
                k, v = [t.strip() for t in ln.split("=", 1)]
                if k == "Count":
                    cur = {"Count": int(v)}
                    cur.update(hdr)
                elif cur is not None:
                    cur[k] = bytes.fromhex(v)
                    if k == "Tag":
                        if (len(cur["Key"]) * 8 == keybits and len(cur["IV"]) == iv_bytes
                                and len(cur["Tag"]) >= 16):
                            cases.append(cur)
                        cur = None
    return cases


def xts_cases(keybits):
    path = os.path.join(GOLDEN, "XTSGenAES%d.rsp" % keybits)
    cases, cur, section = [], None, None
    with open(path) as f:
        for ln in f:
            ln = ln.strip()
            if ln in ("[ENCRYPT]", "[DECRYPT]"):
                section = ln.strip("[]")
            elif "=" in ln:
                k, v = [t.strip() for t in ln.split("=", 1)]
                if k == "COUNT":
                    cur = {"COUNT": int(v), "section": section}
                elif cur is not None:
                    cur[k] = int(v) if k == "DataUnitLen" else bytes.fromhex(v)
                    if "PT" in cur and "CT" in cur:
                        if (len(cur["Key"]) * 4 == keybits
                                and cur["DataUnitLen"] == 8 * len(cur["PT"])):
                            cases.append(cur)
                        cur = None
    return cases


def cmac_cases(keybits):
    path = os.path.join(GOLDEN, "CMACGenAES%d.rsp" % keybits)
    cases, cur = [], {}
    with open(path) as f:
        for ln in f:
            ln = ln.strip()
            if "=" not in ln or ln.startswith("#"):
                continue
            k, v = [t.strip() for t in ln.split("=", 1)]
            if k in ("Count", "Klen", "Mlen", "Tlen"):
                cur[k] = int(v)
            elif k in ("Key", "Msg"):
                cur[k] = bytes.fromhex(v)
            elif k == "Mac":
                cur[k] = bytes.fromhex(v)
                if len(cur["Key"]) * 8 == keybits:
                    if cur["Mlen"] == 0:
                        cur["Msg"] = b""
                    cases.append(cur)
                cur = {}
    return cases


def ccm_cases(keybits, nonce_len=11):
    """VNT<bits>.rsp: the [Nlen = nonce_len] section (the reference's harness takes the one that equals its
    CCM_NONCE_LEN, aes_testvectors_CCM.h:84; every section has 16-byte tags)"""
    path = os.path.join(GOLDEN, "VNT%d.rsp" % keybits)
    cases, key, cur = [], None, {}
    with open(path) as f:
        for ln in f:
            ln = ln.strip()
            if "=" not in ln or ln.startswith("#") or ln.startswith("["):
                continue
            k, v = [t.strip() for t in ln.split("=", 1)]
            if k == "Key":
                key = bytes.fromhex(v)
            elif k == "Count":
                cur = {"Count": int(v), "Key": key}
            elif k in ("Nonce", "Adata", "Payload"):
                cur[k] = bytes.fromhex(v)
            elif k == "CT" and "Payload" in cur:
                cur[k] = bytes.fromhex(v)
                if (len(cur["Key"]) * 8 == keybits and len(cur["Nonce"]) == nonce_len
                        and len(cur["CT"]) - 16 == len(cur["Payload"])):
                    cases.append(cur)
                cur = {}
    return cases


def gcmsiv_cases(keybits):
    """SIV_GCM_ACVP.tv, filter of testvectors/aes_testvectors_GCMSIV.h: key of AES_KEYLENGTH bytes"""
    path = os.path.join(GOLDEN, "SIV_GCM_ACVP.tv")
    cases, cur = [], {}
    with open(path) as f:
        for ln in f:
            ln = ln.strip()
            if "=" not in ln or ln.startswith("#"):
                continue
            k, v = [t.strip() for t in ln.split("=", 1)]
            if k == "Count":
                cur = {"Count": int(v)}
            elif k in ("pt", "key", "aad", "iv", "ct"):
                cur[k] = bytes.fromhex(v)
                if k == "ct":
                    if len(cur["key"]) * 8 == keybits:
                        cases.append(cur)
                    cur = {}
    return cases


def ocb_cases(keybits, nonce_len=12, tag_len=16):
    """OCB_AES128.tv (OpenSSL evp format), filter of testvectors/aes_testvectors_OCB.h:86-90: key of
    AES_KEYLENGTH bytes, OCB_NONCE_LEN-byte nonce, OCB_TAG_LEN-byte tag (the last Tag line of a stanza wins)"""
    path = os.path.join(GOLDEN, "OCB_AES128.tv")
    cases, cur = [], {}

    def flush():
        if {"Key", "IV", "Tag", "Plaintext", "Ciphertext"} <= set(cur) and "Result" not in cur:
            if len(cur["Key"]) * 8 == keybits and len(cur["IV"]) == nonce_len and len(cur["Tag"]) == tag_len:
                cases.append(dict(key=cur["Key"], iv=cur["IV"], aad=cur.get("AAD", b""), pt=cur["Plaintext"],
                                  ct=cur["Ciphertext"] + cur["Tag"]))
    with open(path) as f:
        for ln in f:
            ln = ln.strip()
            if ln.startswith("Cipher ="):
                flush()
                cur = {}
            elif "=" in ln and not ln.startswith("#"):
                k, v = [t.strip() for t in ln.split("=", 1)]
                cur[k] = v if k in ("Result", "Operation") else bytes.fromhex(v)
    flush()
    return cases
