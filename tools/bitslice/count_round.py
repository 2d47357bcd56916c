"""Op count of one bitsliced AES round on 3-input LUTs: S-box alone, and a whole output column
(4 S-boxes -> ShiftRows is wiring -> MixColumns -> AddRoundKey) mapped as ONE netlist."""
import random
import bp_sbox
import lutmap


def sbox_gates(prefix, inbits):
    """inbits: names of input planes, index 0 = bit 0 (LSB).  Returns (gates, outbits LSB-first)."""
    ren = {"U%d" % i: inbits[7 - i] for i in range(8)}
    gates = []
    for dst, op, a, b in bp_sbox.program():
        ren[dst] = prefix + dst
        gates.append((ren[dst], op, ren[a], ren[b]))
    return gates, [ren["S%d" % (7 - i)] for i in range(8)]


def column_gates(last=False, with_key=True):
    """one output column: input bytes a0..a3 (rows 0..3 AFTER ShiftRows), each 8 planes"""
    gates = []
    s = []
    for r in range(4):
        g, o = sbox_gates("b%d_" % r, ["in%d_%d" % (r, i) for i in range(8)])
        gates += g
        s.append(o)
    outs = []
    n = [0]

    def x(a, b):
        n[0] += 1
        d = "x%d" % n[0]
        gates.append((d, "^", a, b))
        return d
    if last:
        for r in range(4):
            for i in range(8):
                outs.append(x(s[r][i], "k%d_%d" % (r, i)) if with_key else s[r][i])
        return gates, outs
    t = [[x(s[r][i], s[(r + 1) % 4][i]) for i in range(8)] for r in range(4)]
    for r in range(4):
        for i in range(8):
            # out[r] = xtime(t[r]) ^ t[r+1] ^ a[r+3]  (a[r+1]^a[r+2] = t[r+1])
            terms = [t[(r + 1) % 4][i], s[(r + 3) % 4][i]]
            terms.append(t[r][i - 1] if i > 0 else t[r][7])
            if i in (1, 3, 4):
                terms.append(t[r][7])
            if with_key:
                terms.append("k%d_%d" % (r, i))
            acc = terms[0]
            for tm in terms[1:]:
                acc = x(acc, tm)
            outs.append(acc)
    return gates, outs


if __name__ == "__main__":
    g, o = sbox_gates("s_", ["i%d" % i for i in range(8)])
    l = lutmap.map_luts(g, o)
    print("S-box alone: %d gates -> %d LUT3" % (len(g), len(l)))
    sb = bp_sbox.aes_sbox()
    for xv in range(256):
        env = lutmap.eval_luts(l, {"i%d" % i: (xv >> i) & 1 for i in range(8)})
        assert sum(env[o[i]] << i for i in range(8)) == sb[xv]
    for last in (False, True):
        g, o = column_gates(last)
        l = lutmap.map_luts(g, o)
        print("column%s: %d gates -> %d LUT3 (x4 columns = %d per round per 32 blocks = %.1f ops/block/round)"
              % (" (last round)" if last else "", len(g), len(l), 4 * len(l), 4 * len(l) / 32.0))
