#!/usr/bin/env python3
"""Generates tools/ubench/issuebench.hip: what does ONE SIMD of a gfx950 CU pay for the instruction mix of a
table-driven AES round -- 16 ds_read_b32 + 12 v_perm_b32 + 12 v_bitop3_b32 per block-round -- when nothing in
the stream depends on anything else?  Every kernel is one `asm volatile` block (explicit registers, the loop
inside the block), 1024 threads = 16 waves per CU, 256 workgroups, conflict-free LDS addresses (lane l reads
bank l mod 32).  The output is cycles per wave-instruction per SIMD (VALU rows) or per CU (LDS rows), from the
slowest wave's s_memtime span: the structural ceiling of the mix, before any data dependency.

Groups:
  R  issue cost of single VALU opcodes (64 independent instructions per loop trip)
  M  16 ds_read_b32 per trip + k VALU instructions, grouped or interleaved
  A  the AES block-round mix in several orders, SGPR or VGPR round keys
  D  ds_read2_b32 / ds_read_u8_d16 in place of ds_read_b32
  W  functional: does the LDS ignore address bits above 17?
  P  does the half-rate / full-rate split survive mixing?
  F  an all-full-rate round (8-byte table entries, no v_perm)
  E  ENERGY (round 5): the block-round mix in its best order (a4r4, VGPR keys) with the byte-extract instruction swapped
     -- v_perm_b32 (SGPR / VGPR selector), v_and_or_b32, v_bfe_u32 + v_lshl_or_b32, SDWA, v_alignbyte, v_lshrrev + and-or --
     to be run with `--sustain S`: every row is launched back to back for S seconds on all 256 CUs so that the power
     manager settles, and the row reports the SETTLED clock and instructions (block-rounds) per second at that clock.
     At the 1.4 kW cap throughput is energy per block, not cycles per block.

    python tools/ubench/gen_issuebench.py && hipcc --offload-arch=gfx950 -O2 -o tools/ubench/issuebench tools/ubench/issuebench.hip
    gpurun -- 'tools/ubench/issuebench [name filter]'          # neither the .hip nor the binary is tracked
    gpurun -- 'tools/ubench/issuebench --sustain 1.5 E_'       # settled clock x instructions/s (energy ranking)
"""
import os

A0, T0, B0, C0 = 40, 56, 72, 88          # address regs, load destinations, VALU chains, constants
NCH = 16
KERNELS = []


def valu(op, i, j=None):
    """instruction text for VALU opcode `op` on chain i (second source: chain j)"""
    d = B0 + (i % NCH)
    s = B0 + ((i + 5) % NCH if j is None else j % NCH)
    c, c2 = C0, C0 + 1
    return {
        "xor":      f"v_xor_b32_e32 v{d}, v{d}, v{s}",
        "and":      f"v_and_b32_e32 v{d}, v{d}, v{s}",
        "or":       f"v_or_b32_e32 v{d}, v{d}, v{s}",
        "mov":      f"v_mov_b32_e32 v{d}, v{s}",
        "lshr16":   f"v_lshrrev_b32_e32 v{d}, 16, v{s}",
        "lshl8":    f"v_lshlrev_b32_e32 v{d}, 8, v{s}",
        "add":      f"v_add_u32_e32 v{d}, v{d}, v{s}",
        "bitop_vvv": f"v_bitop3_b32 v{d}, v{d}, v{s}, v{c} bitop3:0x96",
        "bitop_vvs": f"v_bitop3_b32 v{d}, v{d}, v{s}, s44 bitop3:0x96",
        "andor_vvv": f"v_bitop3_b32 v{d}, v{s}, v{c2}, v{c} bitop3:0xea",
        "perm_vvs": f"v_perm_b32 v{d}, v{d}, v{s}, s45",
        "perm_vvv": f"v_perm_b32 v{d}, v{d}, v{s}, v{c2}",
        "pk_lshl":  f"v_pk_lshlrev_b16 v{d}, v{C0 + 2}, v{s}",
        "and_or":   f"v_and_or_b32 v{d}, v{s}, v{c2}, v{c}",
        "lshl_or":  f"v_lshl_or_b32 v{d}, v{s}, 8, v{c}",
        "bfe":      f"v_bfe_u32 v{d}, v{s}, 8, 8",
        "alignbyte": f"v_alignbyte_b32 v{d}, v{d}, v{s}, 1",
        "alignbit": f"v_alignbit_b32 v{d}, v{d}, v{s}, 8",
        "sdwa_b2":  f"v_or_b32_sdwa v{d}, v{c}, v{s} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2",
    }[op]


def rd(i):
    return f"ds_read_b32 v{T0 + i % NCH}, v{A0 + i % NCH}"


def kernel(name, body, per_trip_valu, per_trip_lds, note="", rand=False):
    KERNELS.append(dict(name=name, body=body, valu=per_trip_valu, lds=per_trip_lds, note=note, rand=rand))


# ---- R: single opcodes
for op in ["xor", "and", "mov", "lshr16", "lshl8", "add", "bitop_vvv", "bitop_vvs", "andor_vvv", "perm_vvs", "perm_vvv",
           "pk_lshl", "and_or", "lshl_or", "bfe", "alignbyte", "alignbit", "sdwa_b2"]:
    kernel("R_" + op, [valu(op, i) for i in range(64)], 64, 0)

# ---- M: 16 reads + k VALU of one kind; g = grouped (VALU first, then the reads), i = interleaved
for op in ["perm_vvs", "bitop_vvv", "xor"]:
    for k in [0, 8, 16, 24, 32, 48]:
        if k == 0 and op != "perm_vvs":
            continue
        grouped = [valu(op, i) for i in range(k)] + [rd(i) for i in range(16)]
        kernel(f"M_{op}_{k}_g", grouped, k, 16)
        if k:
            inter, done = [], 0
            for i in range(16):
                inter.append(rd(i))
                want = (i + 1) * k // 16
                while done < want:
                    inter.append(valu(op, done)); done += 1
            kernel(f"M_{op}_{k}_i", inter, k, 16)


# ---- A: the AES block-round mix: 12 perm + 4 and-or (addresses), 16 reads, 8 xor3 (4 with the round key)
def aes_mix(order, keys):
    addr = []
    for i in range(16):
        addr.append(valu("andor_vvv", i) if i % 4 == 1 else valu("perm_vvs", i))
    xors = []
    for c in range(4):
        xors.append(valu("bitop_vvv", 2 * c))
        xors.append(valu("bitop_vvv" if keys == "v" else "bitop_vvs", 2 * c + 1))
    reads = [rd(i) for i in range(16)]
    if order == "grouped":          # today's order: 16 addresses, 16 reads, 8 combines
        return addr + reads + xors
    if order == "rx":               # combines in the shadow of the reads: 4 reads, 2 xors, ...
        out = list(addr)
        for c in range(4):
            out += reads[4 * c:4 * c + 4] + xors[2 * c:2 * c + 2]
        return out
    if order == "arx":              # everything interleaved: address, read, half a combine
        out = []
        for i in range(16):
            out += [addr[i], reads[i]]
            if i % 2:
                out.append(xors[i // 2])
        return out
    if order == "a4r4":             # four addresses, four reads, two combines
        out = []
        for c in range(4):
            out += addr[4 * c:4 * c + 4] + reads[4 * c:4 * c + 4] + xors[2 * c:2 * c + 2]
        return out
    raise ValueError(order)


for order in ["grouped", "rx", "arx", "a4r4"]:
    for keys in ["s", "v"]:
        kernel(f"A_{order}_k{keys}", aes_mix(order, keys) * 2, 48, 32, "two block-rounds per trip")

# the same with setprio around the issue phase (as the product kernel does)
kernel("A_grouped_ks_prio", (["s_setprio 1"] + aes_mix("grouped", "s")[:32] + ["s_setprio 0"] + aes_mix("grouped", "s")[32:]) * 2, 48, 32)

# ---- D: other DS forms under the same VALU load (grouped order, SGPR keys)
valu24 = aes_mix("grouped", "s")
valu_only = valu24[:16] + valu24[32:]
kernel("D_valu_only", valu_only * 2, 48, 0)
kernel("D_read2", (valu24[:16] + [f"ds_read2_b32 v[{T0 + 2 * i}:{T0 + 2 * i + 1}], v{A0 + i} offset0:0 offset1:32" for i in range(8)] + valu24[32:]) * 2,
       48, 16, "8 ds_read2_b32 = 16 dwords per block-round")
kernel("D_u8_d16", (valu24[:16] + [f"ds_read_u8_d16{'_hi' if i % 2 else ''} v{T0 + i // 2}, v{A0 + i}" for i in range(16)] + valu24[32:]) * 2, 48, 32)
kernel("D_b64", (valu24[:16] + [f"ds_read_b64 v[{T0 + 2 * (i % 8)}:{T0 + 2 * (i % 8) + 1}], v{A0 + i}" for i in range(16)] + valu24[32:]) * 2, 48, 32,
       "addresses 4-byte strided: b64 bank conflicts expected")
kernel("D_reads_only", [rd(i) for i in range(16)] * 2, 0, 32)

# ---- P: does the half-rate / full-rate split survive mixing?  12 v_perm (half rate) + 12 all-VGPR v_bitop3 (full rate)
def pf_mix(run):
    """runs of `run` perms followed by `run` fast ops"""
    out, p, f = [], 0, 0
    while p < 12 or f < 12:
        for _ in range(run):
            if p < 12:
                out.append(valu("perm_vvs", p)); p += 1
        for _ in range(run):
            if f < 12:
                out.append(valu("bitop_vvv", f)); f += 1
    return out


for run in [1, 2, 4, 12]:
    kernel(f"P_valu_run{run}", pf_mix(run) * 2, 48, 0, "12 perm + 12 bitop3(vvv) in runs of %d" % run)
kernel("P_xor_sgpr_e32", [f"v_xor_b32_e32 v{B0 + i % 16}, s44, v{B0 + (i + 5) % 16}" for i in range(64)], 64, 0, "VOP2 with an SGPR source")
kernel("P_fast_pairs_sgpr", sum(([valu("bitop_vvv", 2 * i), valu("bitop_vvs", 2 * i + 1)] for i in range(32)), []), 64, 0, "vvv, vvs alternating")
kernel("P_perm_then_xor", sum(([valu("perm_vvs", 2 * i), valu("xor", 2 * i + 1)] for i in range(32)), []), 64, 0, "perm, xor alternating")
kernel("P_perm2_xor2", sum(([valu("perm_vvs", 4 * i), valu("perm_vvs", 4 * i + 1), valu("xor", 4 * i + 2), valu("xor", 4 * i + 3)] for i in range(16)), []), 64, 0)


# the AES mix with the fast ops adjacent: per slot 3 perm + [1 and-or + 2 xor3(vvv)] ; VGPR keys
def aes_mix_fastgrouped(with_reads):
    out = []
    for c in range(4):
        out += [valu("perm_vvs", 4 * c), valu("perm_vvs", 4 * c + 2), valu("perm_vvs", 4 * c + 3), valu("andor_vvv", 4 * c + 1)]
        if with_reads:
            out += [rd(4 * c + j) for j in range(4)]
        out += [valu("bitop_vvv", 2 * c), valu("bitop_vvv", 2 * c + 1)]
    return out


def aes_mix_fastgrouped2(with_reads):
    """and-or and the two xors back to back BEFORE the perms of the slot"""
    out = []
    for c in range(4):
        out += [valu("bitop_vvv", 2 * c), valu("bitop_vvv", 2 * c + 1), valu("andor_vvv", 4 * c + 1),
                valu("perm_vvs", 4 * c), valu("perm_vvs", 4 * c + 2), valu("perm_vvs", 4 * c + 3)]
        if with_reads:
            out += [rd(4 * c + j) for j in range(4)]
    return out


kernel("P_aes_fg_valu", aes_mix_fastgrouped(False) * 2, 48, 0)
kernel("P_aes_fg_a4r4", aes_mix_fastgrouped(True) * 2, 48, 32)
kernel("P_aes_fg2_valu", aes_mix_fastgrouped2(False) * 2, 48, 0)
kernel("P_aes_fg2_a4r4", aes_mix_fastgrouped2(True) * 2, 48, 32)
# a4r4 with priorities: issue slots at 1, combines at 0
a = aes_mix("a4r4", "v")
kernel("P_a4r4_kv_again", a * 2, 48, 32)
# 8:8 granularity
def a8r8():
    addr = [valu("andor_vvv", i) if i % 4 == 1 else valu("perm_vvs", i) for i in range(16)]
    xors = [valu("bitop_vvv", i) for i in range(8)]
    out = []
    for h in range(2):
        out += addr[8 * h:8 * h + 8] + [rd(8 * h + j) for j in range(8)] + xors[4 * h:4 * h + 4]
    return out
kernel("P_a8r8_kv", a8r8() * 2, 48, 32)
def a2r2():
    addr = [valu("andor_vvv", i) if i % 4 == 1 else valu("perm_vvs", i) for i in range(16)]
    xors = [valu("bitop_vvv", i) for i in range(8)]
    out = []
    for h in range(8):
        out += addr[2 * h:2 * h + 2] + [rd(2 * h + j) for j in range(2)] + xors[h:h + 1]
    return out
kernel("P_a2r2_kv", a2r2() * 2, 48, 32)

# ---- F: an all-full-rate round?  Two state words per column with the index bytes at positions 1 and 3 (8-byte table
# entries, ds_read_b64): byte extraction = and-or (position 1) or v_lshrrev 16 + and-or (position 3), no v_perm at all;
# the price is twice the combines.  Per block-round: 8 shifts + 16 and-or + 16 xor3 = 40 full-rate VALU + 16 ds_read_b64.
def rd64(i):
    return f"ds_read_b64 v[{T0 + 2 * (i % 8)}:{T0 + 2 * (i % 8) + 1}], v{A0 + i % NCH}"


def fast_round(order, slow=0):
    addr = []
    for i in range(16):
        if i % 2:
            addr += [valu("lshr16", i), valu("andor_vvv", i)]
        else:
            addr += [valu("andor_vvv", i)]
    xors = [valu("bitop_vvv", i) for i in range(16)]
    if slow:                                         # a few half-rate instructions sprinkled in
        for k in range(slow):
            xors[(5 * k + 2) % 16] = valu("perm_vvs", 5 * k + 2)
    reads = [rd64(i) for i in range(16)]
    if order == "valu":
        return addr + xors
    if order == "grouped":
        return addr + reads + xors
    if order == "a4r4":
        out = []
        for c in range(4):
            out += addr[6 * c:6 * c + 6] + reads[4 * c:4 * c + 4] + xors[4 * c:4 * c + 4]
        return out
    raise ValueError(order)


kernel("F_fast40_valu", fast_round("valu") * 2, 80, 0)
kernel("F_fast40_valu_2slow", fast_round("valu", 2) * 2, 80, 0, "2 of 40 are v_perm")
kernel("F_fast40_valu_4slow", fast_round("valu", 4) * 2, 80, 0, "4 of 40 are v_perm")
kernel("F_reads64_only", [rd64(i) for i in range(16)] * 2, 0, 32)
kernel("F_fast40_b64_grouped", fast_round("grouped") * 2, 80, 32)
kernel("F_fast40_b64_a4r4", fast_round("a4r4") * 2, 80, 32)
kernel("F_fast40_b64_grouped_2slow", fast_round("grouped", 2) * 2, 80, 32)


# ---- E: energy.  a4r4 order, VGPR keys; `ext` = the instruction(s) that turn one state byte into an LDS address
def energy_mix(ext):
    addr = []
    for i in range(16):
        if i % 4 == 1:
            addr.append([valu("andor_vvv", i)])                 # the byte that already sits in position: and-or today too
        elif ext == "bfe_or":
            addr.append([valu("bfe", i), valu("lshl_or", i)])
        elif ext == "lshr_andor":
            addr.append([valu("lshr16", i), valu("andor_vvv", i)])
        else:
            addr.append([valu(ext, i)])
    xors = [valu("bitop_vvv", i) for i in range(8)]
    out, n = [], 0
    for c in range(4):
        for a in addr[4 * c:4 * c + 4]:
            out += a
            n += len(a)
        out += [rd(4 * c + j) for j in range(4)] + xors[2 * c:2 * c + 2]
        n += 2
    return out, n


for ext in ["perm_vvs", "perm_vvv", "and_or", "andor_vvv", "bfe_or", "sdwa_b2", "alignbyte", "lshr_andor"]:
    body, nv = energy_mix(ext)
    kernel("E_mix_" + ext, body * 2, 2 * nv, 32, "block-round mix, byte extract = %s" % ext)
kernel("E_reads_only", [rd(i) for i in range(16)] * 2, 0, 32, "the 16 lookups of a block-round alone")
for op in ["perm_vvs", "perm_vvv", "and_or", "andor_vvv", "bfe", "lshl_or", "sdwa_b2", "alignbyte", "xor", "bitop_vvv", "bitop_vvs", "lshr16", "mov", "add"]:
    kernel("E_op_" + op, [valu(op, i) for i in range(64)], 64, 0, "single opcode")
# the same streams over RANDOM per-lane data (chains, constants and table rows hashed from the thread id): the streams
# above run on near-constant operands, toggle few wires and never reach the power cap; a cipher's operands are random
for ext in ["perm_vvs", "perm_vvv", "and_or", "andor_vvv", "bfe_or", "sdwa_b2", "alignbyte", "lshr_andor"]:
    body, nv = energy_mix(ext)
    kernel("X_mix_" + ext, body * 2, 2 * nv, 32, "random data; byte extract = %s" % ext, rand=True)
kernel("X_reads_only", [rd(i) for i in range(16)] * 2, 0, 32, "random rows; the 16 lookups alone", rand=True)
for op in ["perm_vvs", "and_or", "andor_vvv", "bfe", "sdwa_b2", "xor", "bitop_vvv", "bitop_vvs", "mov"]:
    kernel("X_op_" + op, [valu(op, i) for i in range(64)], 64, 0, "random data; single opcode", rand=True)

HDR = r'''// GENERATED by tools/ubench/gen_issuebench.py -- do not edit.  Diagnostic, not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef unsigned int u32;
typedef unsigned long long u64;
extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91","s40","s44","s45","scc","memory"
'''

KERN = r'''
__global__ __launch_bounds__(1024) void k_%(name)s(u32 iters, u64 *cycles, u32 *sink)
{
    for (u32 i = threadIdx.x; i < 16384u; i += blockDim.x) ((u32 *)lds)[i] = i * 2654435761u;
    __syncthreads();
    const u32 slot = (threadIdx.x & 31u) * 4u;
    u32 out;
    const u64 t0 = __builtin_readcyclecounter();
    asm volatile(
%(init)s
        "s_mov_b32 s40, %%[iters]\n"
        "s_mov_b32 s44, 0x5a5a5a5a\n"
        "s_mov_b32 s45, 0x0c020500\n"
        "L_%(name)s_%%=:\n"
%(body)s
        "s_sub_u32 s40, s40, 1\n"
        "s_cmp_lg_u32 s40, 0\n"
        "s_cbranch_scc1 L_%(name)s_%%=\n"
        "s_waitcnt lgkmcnt(0)\n"
%(fold)s
        : [out] "=v"(out) : [slot] "v"(slot), [slot8] "v"(slot * 2u), [iters] "s"(iters), [tid] "v"(threadIdx.x + 1024u * blockIdx.x) : CLOB);
    const u64 t1 = __builtin_readcyclecounter();
    if (out == 0x12345678u) sink[0] = out;
    if ((threadIdx.x & 63u) == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
'''


def q(lines):
    return "\n".join('        "%s\\n"' % l for l in lines)


def emit():
    src = [HDR]
    init = []
    for i in range(NCH):
        init.append(f"v_add_u32_e32 v{A0 + i}, {((i * 37 + 11) & 127) * 256}, %[slot]")   # rows differ, bank = lane
        init.append(f"v_mov_b32_e32 v{T0 + i}, 0")
        init.append(f"v_add_u32_e32 v{B0 + i}, {i * 7 + 3}, %[slot]")
    init += [f"v_mov_b32_e32 v{C0}, 0x0000ff00", f"v_mov_b32_e32 v{C0 + 1}, 0x0c020500", f"v_mov_b32_e32 v{C0 + 2}, 0x00080008",
             f"v_mov_b32_e32 v{C0 + 3}, 0"]
    fold = ["v_mov_b32_e32 %[out], 0"]
    for i in range(NCH):
        fold.append(f"v_xor_b32_e32 %[out], %[out], v{T0 + i}")
        fold.append(f"v_xor_b32_e32 %[out], %[out], v{B0 + i}")
    # random per-lane data: chain i = hash(tid, i); constants hashed too; table row of address register i = 7 bits of its hash
    init_rand = []
    for i in range(NCH):
        init_rand += [f"s_mov_b32 s40, {hex((0x9E3779B1 * (2 * i + 1) + 0x7F4A7C15 * i) & 0xffffffff | 1)}",
                      f"v_mul_lo_u32 v{B0 + i}, %[tid], s40",
                      f"v_lshrrev_b32_e32 v{T0 + i}, 15, v{B0 + i}",
                      f"v_xor_b32_e32 v{B0 + i}, v{B0 + i}, v{T0 + i}",
                      f"s_mov_b32 s40, {hex((0x85EBCA6B + 0x27D4EB2F * i) & 0xffffffff | 1)}",
                      f"v_mul_lo_u32 v{B0 + i}, v{B0 + i}, s40",
                      f"v_lshrrev_b32_e32 v{T0 + i}, 13, v{B0 + i}",
                      f"v_xor_b32_e32 v{B0 + i}, v{B0 + i}, v{T0 + i}",
                      f"v_and_b32_e32 v{A0 + i}, 0x7f00, v{B0 + i}",
                      f"v_or_b32_e32 v{A0 + i}, v{A0 + i}, %[slot]",
                      f"v_mov_b32_e32 v{T0 + i}, 0"]
    init_rand += [f"v_mov_b32_e32 v{C0}, 0x0000ff00", f"v_mov_b32_e32 v{C0 + 1}, 0x0c020500", f"v_mov_b32_e32 v{C0 + 2}, 0x00080008",
                  f"v_xor_b32_e32 v{C0 + 3}, v{B0}, v{B0 + 7}"]
    init64 = [l.replace("%[slot]", "%[slot8]") if l.startswith("v_add_u32_e32 v%d" % A0) or any(l.startswith("v_add_u32_e32 v%d," % (A0 + i)) for i in range(NCH)) else l for l in init]
    for k in KERNELS:
        src.append(KERN % dict(name=k["name"], init=q(init_rand if k["rand"] else init64 if k["name"].startswith("F_") else init), body=q(k["body"]), fold=q(fold)))
    # functional test W: address bits above the LDS size
    src.append(r'''
__global__ void k_W_wrap(u32 *res)
{
    for (u32 i = threadIdx.x; i < 16384u; i += blockDim.x) ((u32 *)lds)[i] = i * 2654435761u;
    __syncthreads();
    const u32 a = (threadIdx.x & 63u) * 4u + 1024u;
    u32 v[6];
    asm volatile("ds_read_b32 %0, %6\n ds_read_b32 %1, %7\n ds_read_b32 %2, %8\n ds_read_b32 %3, %9\n ds_read_b32 %4, %10\n ds_read_b32 %5, %11\n s_waitcnt lgkmcnt(0)\n"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
                 : "v"(a), "v"(a | 0x00040000u), "v"(a | 0x00100000u), "v"(a | 0x12340000u), "v"(a | 0x80000000u), "v"(a | 0x00010000u) : "memory");
    for (int i = 0; i < 6; ++i) res[threadIdx.x * 6 + i] = v[i];
}
''')
    src.append(r'''
struct Row { const char *name; void (*fn)(u32, u64 *, u32 *); int valu, ldsn; const char *note; };
static Row rows[] = {
''')
    for k in KERNELS:
        src.append('    {"%s", k_%s, %d, %d, "%s"},\n' % (k["name"], k["name"], k["valu"], k["lds"], k["note"]))
    src.append(r'''};

int main(int argc, char **argv)
{
    const int wgs = 256;
    double sustain = 0;
    const char *only = "";
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--sustain") && i + 1 < argc) sustain = atof(argv[++i]);
        else only = argv[i];
    }
    u64 *d_cyc; u32 *d_sink;
    (void)hipMalloc(&d_cyc, wgs * 16 * sizeof(u64)); (void)hipMalloc(&d_sink, 4096 * 6 * 4);
    u64 *h = (u64 *)malloc(wgs * 16 * sizeof(u64));
    {   // W
        (void)hipFuncSetAttribute((const void *)k_W_wrap, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipLaunchKernelGGL(k_W_wrap, dim3(1), dim3(64), 65536, 0, d_sink);
        u32 r[64 * 6]; (void)hipMemcpy(r, d_sink, sizeof r, hipMemcpyDeviceToHost);
        const char *what[6] = {"plain", "|1<<18", "|1<<20", "|0x1234<<16", "|1<<31", "|1<<16 (64 KiB allocated)"};
        for (int i = 0; i < 6; ++i) {
            int same = 0, zero = 0;
            for (int l = 0; l < 64; ++l) { same += r[l * 6 + i] == r[l * 6]; zero += r[l * 6 + i] == 0; }
            printf("W address %-28s: %2d/64 lanes read the same word as the plain address, %2d read 0\n", what[i], same, zero);
        }
    }
    for (unsigned ri = 0; ri < sizeof rows / sizeof rows[0]; ++ri) {
        const Row &R = rows[ri];
        if (*only && !strstr(R.name, only)) continue;
        const int per = R.valu + R.ldsn;
        const u32 iters = (u32)(4000000 / per);
        (void)hipFuncSetAttribute((const void *)R.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipLaunchKernelGGL(R.fn, dim3(wgs), dim3(1024), 65536, 0, iters / 8, d_cyc, d_sink);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        if (sustain > 0) {
            /* energy ranking: the same launch back to back until `sustain` seconds have passed (the power manager needs
             * far longer than one 20 ms launch), then one more launch timed: its clock is the SETTLED clock of this
             * instruction stream on all 256 CUs, and instructions / its time is what the chip sustains under the cap */
            hipEvent_t s0, s1; (void)hipEventCreate(&s0); (void)hipEventCreate(&s1);
            float run_ms = 0;
            (void)hipEventRecord(s0);
            int launches = 0;
            do {
                for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(R.fn, dim3(wgs), dim3(1024), 65536, 0, iters, d_cyc, d_sink);
                launches += 8;
                (void)hipEventRecord(s1); (void)hipEventSynchronize(s1);
                (void)hipEventElapsedTime(&run_ms, s0, s1);
            } while (run_ms < sustain * 1e3);
            const int NL = 6;                                     // the reported figures: mean of six more launches
            (void)hipEventRecord(e0);
            for (int k = 0; k < NL; ++k) hipLaunchKernelGGL(R.fn, dim3(wgs), dim3(1024), 65536, 0, iters, d_cyc, d_sink);
            (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            ms /= NL;
            (void)hipMemcpy(h, d_cyc, wgs * 16 * sizeof(u64), hipMemcpyDeviceToHost);
            u64 mx = 0;
            for (int i = 0; i < wgs * 16; ++i) if (h[i] > mx) mx = h[i];   // (cycles of the last launch)
            const double trips = 16.0 * iters * wgs;              // wave-trips on the chip
            const double ghz = (double)mx / (ms * 1e6);
            printf("%-22s settled %5.3f GHz after %5.2f s (%3d launches) | last launch %7.3f ms", R.name, ghz, run_ms * 1e-3, launches, ms);
            if (R.valu) printf(" | %7.2f G VALU wave-instr/s", trips * R.valu / (ms * 1e6));
            if (R.ldsn) printf(" | %7.2f G ds_read wave-instr/s | %7.2f G block-rounds/s", trips * R.ldsn / (ms * 1e6), trips * R.ldsn / 16.0 * 64.0 / (ms * 1e6));
            if (R.valu && R.ldsn) printf(" | %6.1f SIMD cyc per 16 reads + %d VALU", 4.0 * (double)mx / (16.0 * iters) * 16.0 / R.ldsn, R.valu * 16 / R.ldsn);
            printf("  %s\n", R.note);
            fflush(stdout);
            continue;
        }
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(R.fn, dim3(wgs), dim3(1024), 65536, 0, iters, d_cyc, d_sink);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h, d_cyc, wgs * 16 * sizeof(u64), hipMemcpyDeviceToHost);
        u64 mx = 0; double avg = 0;
        for (int i = 0; i < wgs * 16; ++i) { if (h[i] > mx) mx = h[i]; avg += (double)h[i]; }
        avg /= wgs * 16;
        const double trips = 16.0 * iters;                     // wave-trips per CU
        printf("%-22s %7.3f ms  %5.2f GHz  wave cyc max %9llu avg %9.0f |", R.name, ms, (double)mx / (ms * 1e6), (unsigned long long)mx, avg);
        if (R.valu) printf(" VALU %5.2f cyc/instr/SIMD", 4.0 * (double)mx / (trips * R.valu));
        if (R.ldsn) printf(" LDS %5.2f clk/instr/CU", (double)mx / (trips * R.ldsn));
        if (R.valu && R.ldsn) printf(" | %6.1f SIMD cyc per 16 reads + %d VALU", 4.0 * (double)mx / trips * 16.0 / R.ldsn, R.valu * 16 / R.ldsn);
        printf("  %s\n", R.note);
        fflush(stdout);
    }
    return 0;
}
''')
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "issuebench.hip")
    with open(path, "w") as f:
        f.write("".join(src))
    print("wrote", path, len(KERNELS), "kernels")


if __name__ == "__main__":
    emit()
