#!/usr/bin/env python3
"""Generates micro-aes_amd/csrc/uaes_rounds_asm.inc.h: the two-block skewed AES round loop (enc_rounds_skewed of
uaes_aes.hip.h; reference rijndaelEncrypt, micro_aes.c:242-259) as ONE hand-scheduled gfx950 inline-asm block per
round count, in several instruction orders selected at compile time by -DUAES_ASM_VARIANT=<n>.

Why by hand: the compiler emits, per block and round, 16 address computations,
then 16 ds_read_b32, then -- after `s_waitcnt lgkmcnt(14)` -- 8 combines.  tools/ubench/issuebench (dependency-free
streams of the same instruction mix, profiles/r04_issuebench.log) shows that this GROUPED order costs 146 SIMD cycles
per block-round against 133 for the LDS alone, while (4 addresses, 4 reads, 2 combines of the OTHER block) x 4 costs
137: the LDS pipe only stays full if every wave keeps offering it reads while its VALU work trickles in between.
The compiler cannot be told to do that (its s_waitcnt placement is per basic block, its scheduler groups by kind),
hence explicit asm with stepped waits.

Register plan of a block X in {A, B}: state SX[0..3] (column words), TX[0..15] (LDS address, then the looked-up
word: `ds_read_b32 v, v`).  Round keys arrive in SGPRs, the lane constants of LaneConst (t[0..3], m1) in VGPRs.

  addr(Y, c)   4 VALU   TY[4c+j] = LDS address of Te_j[ byte j of SY[(c+j)&3] ]      (j = 1: v_bitop3 and-or)
  read(Y, c)   4 DS     TY[4c+j] = lds[TY[4c+j]]
  comb(X, c)   2 VALU   SX[c] = TX[4c] ^ TX[4c+1] ^ TX[4c+2] ^ TX[4c+3] ^ key[c]     (last round: 2 v_perm + or-xor)

lgkmcnt arithmetic: LDS operations return in order and a wave can have at most 15 outstanding.  After read(Y, c) the
reads issued since the last read of X's column c are X's columns c+1..3 and Y's columns 0..c = 16, so
`s_waitcnt lgkmcnt(15)` is exactly "X's column c has arrived" (anything older that is still counted -- the compiler's
own LDS or scalar loads issued before the block -- only makes the wait stronger, never weaker).

Variants:
  1  grouped   16 addr, 16 reads, wait, 8 comb          (the compiler's order: the control)
  2  a4r4      (addr c, read c, wait15, comb c) x 4
  3  a4r4p     variant 2 with s_setprio 1 around addr+read, 0 around comb
  4  rx        16 addr, then (read c, wait, comb c) x 4
  5  a2r2      (2 addr, 2 reads, wait, 1 comb) x 8
  6  a4r4late  (addr c, read c) then comb of column c-1: one slot of slack between a read burst and its use
  7  grouped_d16  variant 1 with the LAST round's sixteen S-box bytes fetched by ds_read_u8_d16 / _d16_hi (byte 1 of
               Te0[x] is S[x]): the four bytes of a column arrive as 00 S2 00 S0 and 00 S3 00 S1, one v_perm interleaves
               them and one v_xor adds the key -- 2 VALU per column instead of 3 (2 v_perm + or-xor)
"""
import os
import sys

VARIANTS = {1: "grouped", 2: "a4r4", 3: "a4r4p", 4: "rx", 5: "a2r2", 6: "a4r4late", 7: "grouped_d16"}
ROUNDS = (8, 10, 12, 14)


class Gen:
    def __init__(self, nrounds, variant, d16=False):
        self.n, self.v, self.d16 = nrounds, variant, d16
        self.lines = []

    # ---- operand names
    def S(self, X, c): return "%%[s%s%d]" % (X.lower(), c & 3)
    def T(self, X, i): return "%%[t%s%d]" % (X.lower(), i)
    def key(self, r, c): return "%%[k%d]" % (4 * (r - 1) + c)

    def emit(self, s): self.lines.append(s)

    def addr(self, Y, c, last, js=range(4)):
        tbl = (2, 3, 0, 1) if last else (0, 1, 2, 3)
        if last and self.d16:
            tbl = (0, 0, 0, 0)
        for j in js:
            src = self.S(Y, c + j)
            dst = self.T(Y, 4 * c + j)
            lt = "%%[lt%d]" % tbl[j]
            if j == 1:
                self.emit("v_bitop3_b32 %s, %s, %%[m1], %s bitop3:0xea" % (dst, src, lt))
            else:
                self.emit("v_perm_b32 %s, %s, %s, %%[sel%d]" % (dst, src, lt, j))

    def read(self, Y, c, js=range(4), last=False):
        if last and self.d16:
            # P = 00 S(b2) 00 S(b0) in SY[c] (the old state word is dead once all sixteen addresses exist -- the
            # grouped order computes them first), Q = 00 S(b3) 00 S(b1) in the d16 scratch register of the column
            P, Q = self.S(Y, c), "%%[q%s%d]" % (Y.lower(), c)
            for j in js:
                dst = P if j in (0, 2) else Q
                self.emit("ds_read_u8_d16%s %s, %s offset:1" % ("_hi" if j >= 2 else "", dst, self.T(Y, 4 * c + j)))
            return
        for j in js:
            t = self.T(Y, 4 * c + j)
            self.emit("ds_read_b32 %s, %s" % (t, t))

    def comb(self, X, c, r, half=None):
        last = r == self.n
        t = [self.T(X, 4 * c + j) for j in range(4)]
        s = self.S(X, c)
        k = self.key(r, c)
        if not last:
            if half in (None, 0):
                self.emit("v_bitop3_b32 %s, %s, %s, %s bitop3:0x96" % (s, t[0], t[1], t[2]))
            if half in (None, 1):
                self.emit("v_bitop3_b32 %s, %s, %s, %s bitop3:0x96" % (s, s, t[3], k))
        elif self.d16:
            q = "%%[q%s%d]" % (X.lower(), c)
            self.emit("v_perm_b32 %s, %s, %s, %%[selD]" % (s, q, s))          # Q.b2 P.b2 Q.b0 P.b0
            self.emit("v_xor_b32_e32 %s, %s, %s" % (s, k, s))
        else:
            if half in (None, 0):
                self.emit("v_perm_b32 %s, %s, %s, %%[selA]" % (t[0], t[1], t[0]))
                self.emit("v_perm_b32 %s, %s, %s, %%[selB]" % (t[2], t[3], t[2]))
            if half in (None, 1):
                self.emit("v_bitop3_b32 %s, %s, %s, %s bitop3:0x56" % (s, t[0], t[2], k))

    def wait(self, n): self.emit("s_waitcnt lgkmcnt(%d)" % n)
    def prio(self, p): self.emit("s_setprio %d" % p)

    # ---- one half-round: Y issues round ry (None: nothing to issue), X combines round rx (None: nothing)
    def phase(self, Y, ry, X, rx):
        v = self.v
        lastY = ry == self.n
        if ry is None:                               # the final combine of the trailing block
            for c in range(4):
                self.wait(4 * (3 - c))
                self.comb(X, c, rx)
            return
        if rx is None:                               # the prologue: the leading block's first lookups
            if v in (1, 4, 7):
                for c in range(4): self.addr(Y, c, lastY)
                for c in range(4): self.read(Y, c, last=lastY)
            else:
                for c in range(4):
                    self.addr(Y, c, lastY); self.read(Y, c)
            return
        if v in (1, 7):
            self.prio(1)
            for c in range(4): self.addr(Y, c, lastY)
            for c in range(4): self.read(Y, c, last=lastY)
            self.prio(0)
            self.wait(15)                            # all of X has arrived once <= 16 are outstanding; 15 is encodable
            for c in range(4): self.comb(X, c, rx)
        elif v in (2, 3):
            for c in range(4):
                if v == 3: self.prio(1)
                self.addr(Y, c, lastY); self.read(Y, c)
                if v == 3: self.prio(0)
                self.wait(15)
                self.comb(X, c, rx)
        elif v == 4:
            for c in range(4): self.addr(Y, c, lastY)
            for c in range(4):
                self.read(Y, c)
                self.wait(15)
                self.comb(X, c, rx)
        elif v == 5:
            for c in range(4):
                for h in range(2):
                    self.addr(Y, c, lastY, js=(2 * h, 2 * h + 1)); self.read(Y, c, js=(2 * h, 2 * h + 1))
                    # issued since X's read 4c+2 (h = 0) / 4c+3 (h = 1): 15 / 16 -> lgkmcnt(15) in both cases
                    self.wait(15)
                    self.comb(X, c, rx, half=h)
        elif v == 6:
            for c in range(4):
                self.addr(Y, c, lastY); self.read(Y, c)
                if c:
                    self.wait(15)                    # stronger than needed (column c-1 needs <= 20): free, the window is 15
                    self.comb(X, c - 1, rx)
            self.wait(15)                            # X's column 3: exactly Y's 16 reads follow it
            self.comb(X, 3, rx)
        else:
            raise ValueError(v)

    def body(self):
        n = self.n
        self.phase("A", 1, None, None)
        for r in range(1, n + 1):
            self.phase("B", r, "A", r)
            if r < n:
                self.phase("A", r + 1, "B", r)
            else:
                self.phase(None, None, "B", r)
        return self.lines


def function(nrounds, variant):
    g = Gen(nrounds, variant, d16=variant == 7)
    body = g.body()
    out = []
    out.append("template <> __device__ __forceinline__ void enc_rounds_asm<%d>(u32 (&sa)[4], u32 (&sb)[4], const u32 *rk, const LaneConst &lc)" % nrounds)
    out.append("{")
    out.append("    u32 " + ", ".join("ta%d" % i for i in range(16)) + ";")
    out.append("    u32 " + ", ".join("tb%d" % i for i in range(16)) + ";")
    if g.d16:
        out.append("    u32 qa0, qa1, qa2, qa3, qb0, qb1, qb2, qb3;")
    out.append("    asm volatile(")
    for l in body:
        out.append('        "%s\\n"' % l)
    outs = ['[sa%d] "+v"(sa[%d])' % (i, i) for i in range(4)] + ['[sb%d] "+v"(sb[%d])' % (i, i) for i in range(4)]
    outs += ['[ta%d] "=&v"(ta%d)' % (i, i) for i in range(16)] + ['[tb%d] "=&v"(tb%d)' % (i, i) for i in range(16)]
    if g.d16:
        outs += ['[q%s%d] "=&v"(q%s%d)' % (x, i, x, i) for x in "ab" for i in range(4)]
    ins = ['[lt%d] "v"(lc.t[%d])' % (i, i) for i in range(4)] + ['[m1] "v"(lc.m1)']
    ins += ['[sel0] "s"(0x0c020400u)', '[sel2] "s"(0x0c020600u)', '[sel3] "s"(0x0c020700u)',
            '[selA] "s"(0x0c0c0500u)', '[selB] "s"(0x07020c0cu)', '[selD] "s"(0x06020400u)']
    ins += ['[k%d] "s"(rk[%d])' % (i, i) for i in range(4 * nrounds)]
    out.append("        : " + ", ".join(outs))
    out.append("        : " + ", ".join(ins) + ");")
    out.append("}")
    return "\n".join(out)


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "micro-aes_amd", "csrc", "uaes_rounds_asm.inc.h")
    src = ["/* GENERATED by tools/gen_rounds_asm.py -- do not edit; see that file for the schedule and the lgkmcnt arithmetic.",
           " * enc_rounds_asm<N>: rounds 1..N (N = the last, MixColumns-free round) of two blocks half a round out of phase;",
           " * rk = the N round keys that follow the state's last AddRoundKey (wave-uniform, SGPRs).  Included by uaes_aes.hip.h. */",
           "template <int NROUNDS> __device__ __forceinline__ void enc_rounds_asm(u32 (&sa)[4], u32 (&sb)[4], const u32 *rk, const LaneConst &lc);"]
    for v in sorted(VARIANTS):
        src.append("#if UAES_ASM_VARIANT == %d   /* %s */" % (v, VARIANTS[v]))
        for n in ROUNDS:
            src.append(function(n, v))
        src.append("#endif")
    with open(path, "w") as f:
        f.write("\n".join(src) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
