"""The Boyar-Peralta depth-16 / 113-gate style straight-line program for the AES S-box
(J. Boyar, R. Peralta, "A new combinational logic minimization technique with applications to
cryptology", SEA 2010), as a list of (dst, op, a, b) with op in {"^", "&", "#"} (# = XNOR).
U0 = most significant input bit ... U7 = least; S0 = most significant output bit.
Checked exhaustively against the S-box derived from GF(2^8) arithmetic (verify())."""

TOP = """
T1 = U0 ^ U3; T2 = U0 ^ U5; T3 = U0 ^ U6; T4 = U3 ^ U5; T5 = U4 ^ U6; T6 = T1 ^ T5; T7 = U1 ^ U2;
T8 = U7 ^ T6; T9 = U7 ^ T7; T10 = T6 ^ T7; T11 = U1 ^ U5; T12 = U2 ^ U5; T13 = T3 ^ T4; T14 = T6 ^ T11;
T15 = T5 ^ T11; T16 = T5 ^ T12; T17 = T9 ^ T16; T18 = U3 ^ U7; T19 = T7 ^ T18; T20 = T1 ^ T19;
T21 = U6 ^ U7; T22 = T7 ^ T21; T23 = T2 ^ T22; T24 = T2 ^ T10; T25 = T20 ^ T17; T26 = T3 ^ T16; T27 = T1 ^ T12;
"""
MID = """
M1 = T13 & T6; M2 = T23 & T8; M3 = T14 ^ M1; M4 = T19 & U7; M5 = M4 ^ M1; M6 = T3 & T16; M7 = T22 & T9;
M8 = T26 ^ M6; M9 = T20 & T17; M10 = M9 ^ M6; M11 = T1 & T15; M12 = T4 & T27; M13 = M12 ^ M11; M14 = T2 & T10;
M15 = M14 ^ M11; M16 = M3 ^ M2; M17 = M5 ^ T24; M18 = M8 ^ M7; M19 = M10 ^ M15; M20 = M16 ^ M13; M21 = M17 ^ M15;
M22 = M18 ^ M13; M23 = M19 ^ T25; M24 = M22 ^ M23; M25 = M22 & M20; M26 = M21 ^ M25; M27 = M20 ^ M21;
M28 = M23 ^ M25; M29 = M28 & M27; M30 = M26 & M24; M31 = M20 & M23; M32 = M27 & M31; M33 = M27 ^ M25;
M34 = M21 & M22; M35 = M24 & M34; M36 = M24 ^ M25; M37 = M21 ^ M29; M38 = M32 ^ M33; M39 = M23 ^ M30;
M40 = M35 ^ M36; M41 = M38 ^ M40; M42 = M37 ^ M39; M43 = M37 ^ M38; M44 = M39 ^ M40; M45 = M42 ^ M41;
M46 = M44 & T6; M47 = M40 & T8; M48 = M39 & U7; M49 = M43 & T16; M50 = M38 & T9; M51 = M37 & T17;
M52 = M42 & T15; M53 = M45 & T27; M54 = M41 & T10; M55 = M44 & T13; M56 = M40 & T23; M57 = M39 & T19;
M58 = M43 & T3; M59 = M38 & T22; M60 = M37 & T20; M61 = M42 & T1; M62 = M45 & T4; M63 = M41 & T2;
"""
BOT = """
L0 = M61 ^ M62; L1 = M50 ^ M56; L2 = M46 ^ M48; L3 = M47 ^ M55; L4 = M54 ^ M58; L5 = M49 ^ M61; L6 = M62 ^ L5;
L7 = M46 ^ L3; L8 = M51 ^ M59; L9 = M52 ^ M53; L10 = M53 ^ L4; L11 = M60 ^ L2; L12 = M48 ^ M51; L13 = M50 ^ L0;
L14 = M52 ^ M61; L15 = M55 ^ L1; L16 = M56 ^ L0; L17 = M57 ^ L1; L18 = M58 ^ L8; L19 = M63 ^ L4; L20 = L0 ^ L1;
L21 = L1 ^ L7; L22 = L3 ^ L12; L23 = L18 ^ L2; L24 = L15 ^ L9; L25 = L6 ^ L10; L26 = L7 ^ L9; L27 = L8 ^ L10;
L28 = L11 ^ L14; L29 = L11 ^ L17;
S0 = L6 ^ L24; S1 = L16 # L26; S2 = L19 # L28; S3 = L6 ^ L21; S4 = L20 ^ L22; S5 = L25 ^ L29; S6 = L13 # L27; S7 = L6 # L23;
"""


def program():
    prog = []
    for stmt in (TOP + MID + BOT).replace("\n", " ").split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        dst, rhs = [x.strip() for x in stmt.split("=")]
        a, op, b = rhs.split()
        prog.append((dst, op, a, b))
    return prog


def aes_sbox():
    def mul(a, b):
        r = 0
        while b:
            if b & 1:
                r ^= a
            a = ((a << 1) ^ (0x11b if a & 0x80 else 0)) & 0x1ff
            b >>= 1
        return r & 0xff
    inv = [0] * 256
    for a in range(1, 256):
        for b in range(1, 256):
            if mul(a, b) == 1:
                inv[a] = b
    sb = []
    for x in range(256):
        v = inv[x]
        s = v
        for k in range(1, 5):
            s ^= ((v << k) | (v >> (8 - k))) & 0xff
        sb.append(s ^ 0x63)
    return sb


def evaluate(prog, x):
    env = {"U%d" % i: (x >> (7 - i)) & 1 for i in range(8)}
    for dst, op, a, b in prog:
        va, vb = env[a], env[b]
        env[dst] = (va ^ vb) if op == "^" else (va & vb) if op == "&" else (1 ^ va ^ vb)
    return sum(env["S%d" % i] << (7 - i) for i in range(8))


def verify():
    prog, sb = program(), aes_sbox()
    bad = [x for x in range(256) if evaluate(prog, x) != sb[x]]
    return len(prog), bad


if __name__ == "__main__":
    n, bad = verify()
    print("gates:", n, "mismatches:", len(bad), bad[:8])
