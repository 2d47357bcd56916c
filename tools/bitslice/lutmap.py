"""Technology mapping of a 2-input XOR/AND/XNOR netlist onto 3-input LUTs (v_bitop3_b32).

Netlist: list of (dst, op, a, b); op in "^", "&", "#"(xnor).  Signals not defined by a gate are
primary inputs.  Mapping = classic K-cut enumeration (K = 3) + area-flow selection + cover from the
outputs, followed by a few passes of exact-area recovery.  Returns the LUT list
[(dst, (leaf0, leaf1, leaf2 or fewer), truth_table)] in topological order; truth table bit
(a<<2 | b<<1 | c) is the function value for leaf values a, b, c (leaf0 = a ...), which is the
immediate of v_bitop3_b32 when leaf0 -> src0 (0xF0), leaf1 -> src1 (0xCC), leaf2 -> src2 (0xAA)."""
import itertools

K = 3


def topo(gates):
    return gates  # generators emit in topological order


def map_luts(gates, outputs, max_cuts=12, passes=3):
    node = {g[0]: g for g in gates}
    order = [g[0] for g in gates]
    fanout = {}
    for d, op, a, b in gates:
        fanout[a] = fanout.get(a, 0) + 1
        fanout[b] = fanout.get(b, 0) + 1
    for o in outputs:
        fanout[o] = fanout.get(o, 0) + 1
    cuts = {}

    def cuts_of(s):
        return cuts[s] if s in node else [frozenset([s])]

    af = {}
    best = {}

    def aflow(s):
        return af.get(s, 0.0)

    for d in order:
        _, op, a, b = node[d]
        cs = set()
        for ca in cuts_of(a):
            for cb in cuts_of(b):
                c = ca | cb
                if len(c) <= K:
                    cs.add(c)
        scored = []
        for c in cs:
            cost = 1.0 + sum(aflow(l) / max(1, fanout.get(l, 1)) for l in c)
            scored.append((cost, len(c), sorted(c), c))
        scored.sort(key=lambda t: (t[0], t[1], t[2]))
        scored = scored[:max_cuts]
        best[d] = scored[0][3]
        af[d] = scored[0][0]
        cuts[d] = [t[3] for t in scored] + [frozenset([d])]

    def cover(choice):
        used, stack = set(), list(outputs)
        while stack:
            s = stack.pop()
            if s in used or s not in node:
                continue
            used.add(s)
            stack.extend(choice[s])
        return used

    choice = dict(best)
    used = cover(choice)
    # exact-area recovery: for each used node try every cut and keep the one that minimises the
    # number of LUTs in the cover (reference counting)
    for _ in range(passes):
        refs = {}
        for s in used:
            for l in choice[s]:
                refs[l] = refs.get(l, 0) + 1
        for o in outputs:
            refs[o] = refs.get(o, 0) + 1

        def deref(s):
            """remove s's LUT: returns number of LUTs freed"""
            n = 1
            for l in choice[s]:
                refs[l] -= 1
                if refs[l] == 0 and l in node:
                    n += deref(l)
            return n

        def ref(s, c):
            n = 1
            for l in c:
                refs[l] = refs.get(l, 0) + 1
                if refs[l] == 1 and l in node:
                    n += ref(l, choice[l])
            return n

        for d in reversed(order):
            if refs.get(d, 0) == 0:
                continue
            cur = choice[d]
            deref(d)
            best_c, best_n = None, None
            for c in cuts[d][:-1]:
                n = ref(d, c)
                # undo
                choice_backup = choice[d]
                choice[d] = c
                deref(d)
                choice[d] = choice_backup
                if best_n is None or n < best_n:
                    best_c, best_n = c, n
            choice[d] = best_c
            ref(d, best_c)
        used = cover(choice)

    # truth tables
    luts = []
    for d in order:
        if d not in used:
            continue
        leaves = sorted(choice[d], key=lambda s: (s not in node, s))
        leaves = list(leaves)
        tt = 0
        for idx in range(1 << len(leaves)):
            env = {}
            for i, l in enumerate(leaves):
                env[l] = (idx >> (len(leaves) - 1 - i)) & 1

            def ev(s):
                if s in env:
                    return env[s]
                _, op, a, b = node[s]
                va, vb = ev(a), ev(b)
                r = (va ^ vb) if op == "^" else (va & vb) if op == "&" else (1 ^ va ^ vb)
                env[s] = r
                return r
            if ev(d):
                tt |= 1 << idx
        luts.append((d, leaves, tt))
    return luts


def eval_luts(luts, inputs):
    env = dict(inputs)
    for d, leaves, tt in luts:
        idx = 0
        for l in leaves:
            idx = (idx << 1) | env[l]
        env[d] = (tt >> idx) & 1
    return env
