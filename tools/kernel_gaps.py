#!/usr/bin/env python3
"""tools/kernel_gaps.py <dir with a rocprofv3 --kernel-trace csv> [n] -- the last n launches of the trace: kernel,
duration and the gap to the previous kernel's end (us).  What a multi-launch call is made of."""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ks = [(r['Kernel_Name'].split('(')[0][:44], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
ks = [k for k in ks if k[0].startswith(('k_', 'void k_'))]
for i in range(len(ks) - n, len(ks)):
    k, p = ks[i], ks[i - 1]
    print("  %-46s %8.2f  gap %7.2f" % (k[0], (k[2] - k[1]) / 1e3, (k[1] - p[2]) / 1e3))
