#!/usr/bin/env python
"""GPU time line of one steady-state tracking frame from a rocprofv3 --kernel-trace database of tools/track_probe.py: the kernels of the
last liw_solve (from its first k_lin_all to k_pack_result), their durations and the idle time between them.
Usage: track_timeline.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s
                          on d.kernel_id = s.id order by d.start"""))
short = lambda n: n.split("liw")[-1][:28]
packs = [i for i, r in enumerate(rows) if "k_pack_result" in r[0]]
for which in (-1, -3):
    i1 = packs[which]
    i0 = packs[which - 1] + 1
    t0 = rows[i0][1]
    busy = 0
    print("frame ending at dispatch %d:" % i1)
    for i in range(i0, i1 + 1):
        n, a, b = rows[i]
        gap = (a - rows[i - 1][2]) / 1e3 if i > i0 else 0.0
        busy += b - a
        print("  %-30s start %7.1f us  dur %6.1f us  gap before %5.1f us" % (short(n), (a - t0) / 1e3, (b - a) / 1e3, gap))
    print("  span %.1f us, busy %.1f us" % ((rows[i1][2] - t0) / 1e3, busy / 1e3))
