"""Parsers for the NIST CAVP .rsp fixtures, applying the SAME case filters as
the reference's harness so that the case counts (375 / 800 / 600) are
themselves a parity check on the parser:

* GCM  -- testvectors/aes_testvectors_GCM.h:86: run a case only if
          Keylen == AES_KEYLENGTH, IVlen == GCM_NONCE_LEN (12 bytes) and
          Taglen >= GCM_TAG_LEN (16 bytes).
* XTS  -- testvectors/aes_testvectors_XTS.h:84: run a case only if the key is
          2*AES_KEYLENGTH bytes and DataUnitLen == 8 * len(PT) (whole bytes).
"""
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gcm_cases(keybits):
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
                        if (len(cur["Key"]) * 8 == keybits and len(cur["IV"]) == 12
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
