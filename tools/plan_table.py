#!/usr/bin/env python3
"""Print the table of arrangements (csrc/uaes_plan.h) as this device -- or, without one, a 256-CU MI355X -- decides it:
every boundary at which the arrangement (or its GHASH positions per thread) changes, found by bisection over
uaes_debug_plan().  DESIGN.md section 3 quotes this output."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import micro_aes_amd as uaes

MIB = 1 << 20


def pts(lo, hi, unit):
    ks, j = {0, 1, 2, 3, 5, 7}, 1
    while (1 << j) * unit <= hi * 2:
        ks.update({(1 << j) - 1, 1 << j, (1 << j) + 1, 3 << (j - 1)})
        j += 1
    return sorted(k * unit for k in ks if lo <= k * unit <= hi)


def walk(title, fn, lo, hi, unit=16, fmt=lambda n: "%11d B (%9.3f MiB)" % (n, n / MIB)):
    print(title)
    p = pts(lo, hi, unit)
    cur = fn(p[0])
    print("    from %s  %-13s launches %d%s" % (fmt(p[0]), cur[0], cur[1], ("  positions/thread %d" % cur[3]) if cur[3] else ""))
    for left, right in zip(p, p[1:]):
        rl = fn(left)
        while (rl[0], rl[3]) != (fn(right)[0], fn(right)[3]):
            a, b = left, right
            while b - a > unit:
                m = (a + b) // 2 // unit * unit
                if (fn(m)[0], fn(m)[3]) == (rl[0], rl[3]):
                    a = m
                else:
                    b = m
            nb = fn(b)
            print("    from %s  %-13s launches %d%s" % (fmt(b), nb[0], nb[1], ("  positions/thread %d" % nb[3]) if nb[3] else ""))
            left, rl = b, nb


walk("ECB", lambda n: uaes.plan("ecb", n), 0, 64 * MIB)
walk("CTR (56-bit big-endian counter)", lambda n: uaes.plan("ctr", n), 0, 64 * MIB)
walk("XTS, one data unit of n bytes", lambda n: uaes.plan("xts", max(n, 16), 1), 16, 64 * MIB)
for sector in (512, 4096):
    walk("XTS, k data units of %d bytes" % sector, lambda k: uaes.plan("xts", sector, max(k, 1)), 1, (8 << 30) // sector, 1,
         lambda k: "%9d units (%9.3f MiB)" % (k, k * sector / MIB))
walk("GCM encrypt", lambda n: uaes.plan("gcm", n), 0, 2048 * MIB)
walk("GCM decrypt, tag first (default, N7)", lambda n: uaes.plan("gcm", n, 0, 1), 0, 2048 * MIB)
walk("GCM decrypt, one pass (uaes_set_gcm_one_pass_decrypt)", lambda n: uaes.plan("gcm", n, 0, 2), 0, 2048 * MIB)
walk("OCB", lambda n: uaes.plan("ocb", n), 0, 64 * MIB)
walk("GCM-SIV", lambda n: uaes.plan("siv", n), 0, 2048 * MIB)
