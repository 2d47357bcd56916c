#!/usr/bin/env python3
"""Golden vectors for the reference's OTHER compile-time length constants (micro_aes.h:103-116): outputs of
the compiled reference built with CCM_NONCE_LEN / CCM_TAG_LEN / GCM_TAG_LEN / OCB_NONCE_LEN / OCB_TAG_LEN patched
(oracle/Makefile: libmicroaes_ref_128_lensA.so, libmicroaes_ref_256_lensB.so; the table is Reference.LENS).
Run in the build container (needs oracle/_ref):  python tests/golden/make_lens_fixtures.py
Writes tests/golden/lens_vectors.json -- inputs and expected outputs only."""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.pyoracle import Reference  # noqa: E402

rnd = random.Random(20260929)
out = {}
for name, spec in sorted(Reference.LENS.items()):
    bits = spec[0]
    ref = Reference(bits, lens=name)
    cases = []
    for n, alen in ((0, 0), (0, 20), (5, 0), (16, 16), (33, 7), (100, 41), (257, 300), (4096, 13)):
        key, aad, pt = rnd.randbytes(bits // 8), rnd.randbytes(alen), rnd.randbytes(n)
        c = {"key": key.hex(), "aad": aad.hex(), "pt": pt.hex()}
        for mode, nlen, enc in (("gcm", 12, ref.gcm_encrypt), ("ccm", ref.ccm_nonce, ref.ccm_encrypt),
                                ("ocb", ref.ocb_nonce, ref.ocb_encrypt)):
            nonce = rnd.randbytes(nlen)
            c[mode] = {"nonce": nonce.hex(), "out": enc(key, nonce, aad, pt).hex()}
        cases.append(c)
    out[name] = {"bits": bits, "ccm_nonce": ref.ccm_nonce, "ccm_tag": ref.ccm_tag, "gcm_tag": ref.gcm_tag,
                 "ocb_nonce": ref.ocb_nonce, "ocb_tag": ref.ocb_tag, "cases": cases}
with open(os.path.join(HERE, "lens_vectors.json"), "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
print("wrote lens_vectors.json:", {k: len(v["cases"]) for k, v in out.items()})
