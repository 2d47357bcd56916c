#!/usr/bin/env python3
"""Registers, LDS and private-segment (scratch) size of every kernel in libuaes_hip.so, read from the
code objects' metadata notes (llvm-objdump --offloading + llvm-readelf --notes; needs no GPU).
Usage: kernel_resources.py [--scratch-only]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib=None):
    lib = lib or os.path.join(ROOT, "micro-aes_amd", "lib", "libuaes_hip.so")
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            assert f.endswith("gfx950"), "code object for another target: " + f
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)],
                                   check=True, capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
                get = lambda k: re.search(r"\.%s:\s*(\S+)" % k, blk).group(1)
                name = subprocess.run(["c++filt", get("name")], capture_output=True,
                                      text=True).stdout.strip()
                out.append(dict(name=name, vgpr=int(get("vgpr_count")), sgpr=int(get("sgpr_count")),
                                lds=int(get("group_segment_fixed_size")), scratch=int(get("private_segment_fixed_size")),
                                vgpr_spill=int(get("vgpr_spill_count")), sgpr_spill=int(get("sgpr_spill_count"))))
    return out


if __name__ == "__main__":
    ks = kernels()
    for k in sorted(ks, key=lambda k: k["name"]):
        if "--scratch-only" in sys.argv and not (k["scratch"] or k["vgpr_spill"]):
            continue
        print("%-110s vgpr %3d sgpr %3d lds %6d scratch %4d spills v%d s%d" % (
            k["name"][:110], k["vgpr"], k["sgpr"], k["lds"], k["scratch"], k["vgpr_spill"], k["sgpr_spill"]))
    print(len(ks), "kernels")
