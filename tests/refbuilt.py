"""The reference-built evidence must not vanish silently.

oracle/_ref/ holds what oracle/Makefile builds from /root/reference with plain gcc: the reference itself as shared
libraries (the pinned checker and bench.py's `cpu_baseline`, kind "reference") and the reference's own callers --
main.c and the testvectors/ harness -- linked to libmicro_aes_hip_<bits>.so (the drop-in proof).  The directory is
git-ignored and reaches the GPU box only because gpurun ships untracked files.  A test that needs one of these files
therefore FAILS when it is missing; only UAES_ALLOW_NO_REF=1 turns that into a skip (a checkout without
/root/reference and without the prebuilt directory).  The list of files is read from oracle/Makefile itself."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def allow_missing():
    return os.environ.get("UAES_ALLOW_NO_REF", "") not in ("", "0")


def makefile_targets():
    """every oracle/_ref/<name> the `ref` and `dropin` targets of oracle/Makefile name as prerequisites, plus the
    main_hip_<bits> binaries the harness_hip_% rule writes beside its target"""
    with open(os.path.join(ROOT, "oracle", "Makefile")) as f:
        text = f.read().replace("\\\n", " ")
    names = []
    for line in text.splitlines():
        if re.match(r"^(ref|dropin):\s*\$\(HERE\)_ref/", line):
            names += re.findall(r"\$\(HERE\)_ref/([\w.]+)", line)
    names += ["main_hip_%s" % n.rsplit("_", 1)[1] for n in names if re.fullmatch(r"harness_hip_\d+", n)]
    return sorted(set(names))


def need(name):
    """path of oracle/_ref/<name>; fails the test (or skips under UAES_ALLOW_NO_REF=1) when it is not there"""
    path = os.path.join(REF_DIR, name)
    if not os.path.exists(path):
        missing("oracle/_ref/%s" % name)
    return path


def missing(what):
    msg = ("%s is missing: the reference-built evidence did not travel / was not built (make -C oracle with "
           "/root/reference present).  Set UAES_ALLOW_NO_REF=1 to skip instead." % what)
    if allow_missing():
        pytest.skip(msg)
    pytest.fail(msg)
