#!/usr/bin/env python3
"""Registers, LDS and private-segment (scratch) size of every kernel in libuaes_hip.so, read from the
code objects' metadata notes (llvm-objdump --offloading + llvm-readelf --notes; needs no GPU).
Usage: kernel_resources.py [--scratch-only]
       kernel_resources.py --disasm <kernel-name-substring> ... [--out DIR]
           the hot loop of each kernel (the backward-branch loop holding the most ds_read_b32) from llvm-objdump -d of
           the shipped code object: instruction counts per loop trip by opcode and the listing itself, so that DESIGN.md's
           "VALU + LDS instructions per block" can be re-derived from a tracked file (profiles/r04_isa/)."""
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


def code_objects(lib=None):
    """(temporary directory, [paths of the gfx950 code objects]) extracted from the library"""
    lib = lib or os.path.join(ROOT, "micro-aes_amd", "lib", "libuaes_hip.so")
    tmp = tempfile.mkdtemp(prefix="uaes_co_")
    so = os.path.join(tmp, "lib.so")
    shutil.copy(lib, so)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, capture_output=True)
    return tmp, [os.path.join(tmp, f) for f in sorted(os.listdir(tmp)) if "amdgcn" in f]


def disassemble(lib=None):
    """{demangled kernel name: [(address, mnemonic, operands)]} for every kernel of the library"""
    tmp, cos = code_objects(lib)
    out = {}
    try:
        for co in cos:
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True,
                                 capture_output=True, text=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                    cur = out.setdefault(name, [])
                    continue
                m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
                if m and cur is not None:
                    cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def hot_loop(insts):
    """the backward-branch loop with the most ds_read_b32: (start index, end index) into insts"""
    addr_ix = {a: i for i, (a, _, _) in enumerate(insts)}
    loops = []
    for i, (a, op, args) in enumerate(insts):
        if not op.startswith("s_cbranch"):
            continue
        m2 = re.match(r"^(\d+)", args.strip())              # the simm16, printed unsigned: target = pc + 4 + 4 * simm16
        if not m2:
            continue
        off = int(m2.group(1))
        tgt = a + 4 + 4 * (off - 65536 if off >= 32768 else off)
        if tgt not in addr_ix or tgt >= a:
            continue
        j = addr_ix[tgt]
        loops.append((i - j + 1, sum(1 for (_, o, _) in insts[j:i + 1] if o == "ds_read_b32"), j, i))
    if not loops:
        return None
    # the loop with the most lookups that is not the shell around another one: skip a loop if one strictly inside it
    # holds a round's worth of lookups itself (the chunk loop of the shared-round CTR kernels wraps the iteration loop,
    # its refill and the odd last iteration)
    for size, n, j, i in sorted(loops, key=lambda l: (-l[1], l[0])):
        inner = [l for l in loops if l[2] >= j and l[3] <= i and (l[2], l[3]) != (j, i)]
        if not any(l[1] >= 64 for l in inner):
            return (j, i)
    return None


def loop_report(name, insts):
    lp = hot_loop(insts)
    if lp is None:
        return "%s: no backward branch found\n" % name
    body = insts[lp[0]:lp[1] + 1]
    counts = {}
    for _, op, _ in body:
        counts[op] = counts.get(op, 0) + 1
    valu = sum(n for o, n in counts.items() if o.startswith("v_"))
    lds = sum(n for o, n in counts.items() if o.startswith("ds_"))
    lines = ["# %s" % name, "# hot loop: %d instructions, %d VALU, %d DS, %d SALU/other per trip" %
             (len(body), valu, lds, len(body) - valu - lds),
             "# by opcode: " + ", ".join("%s %d" % (o, n) for o, n in sorted(counts.items(), key=lambda x: -x[1]))]
    lines += ["  %06x  %-24s %s" % (a, op, args) for a, op, args in body]
    return "\n".join(lines) + "\n", counts


if __name__ == "__main__":
    if "--disasm" in sys.argv:
        args = sys.argv[sys.argv.index("--disasm") + 1:]
        out_dir = None
        if "--out" in args:
            out_dir = args[args.index("--out") + 1]
            args = args[:args.index("--out")]
            os.makedirs(out_dir, exist_ok=True)
        dis = disassemble()
        for want in args:
            for name, insts in sorted(dis.items()):
                if want not in name:
                    continue
                rep = loop_report(name, insts)
                text = rep if isinstance(rep, str) else rep[0]
                if out_dir:
                    fn = re.sub(r"[^A-Za-z0-9_]+", "_", name.split("(")[0]).strip("_") + ".hotloop.txt"
                    with open(os.path.join(out_dir, fn), "w") as f:
                        f.write(text)
                    print(text.splitlines()[0], "->", fn)
                    print("\n".join(text.splitlines()[1:3]))
                else:
                    print(text)
        sys.exit(0)
    ks = kernels()
    for k in sorted(ks, key=lambda k: k["name"]):
        if "--scratch-only" in sys.argv and not (k["scratch"] or k["vgpr_spill"]):
            continue
        print("%-110s vgpr %3d sgpr %3d lds %6d scratch %4d spills v%d s%d" % (
            k["name"][:110], k["vgpr"], k["sgpr"], k["lds"], k["scratch"], k["vgpr_spill"], k["sgpr_spill"]))
    print(len(ks), "kernels")
