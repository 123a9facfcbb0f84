#!/usr/bin/env python
"""Time line of the batched LM loop from a rocprofv3 --kernel-trace database (rocpd sqlite): for every k_lm_step_quad launch of the
timed solve, when the kernels of the iteration behind it start and end relative to the END of that step kernel, and how long the device
sits idle between the last kernel of one iteration and the step kernel of the next.  Usage: rocprof_timeline.py results.db [first] [count]"""
import sqlite3
import sys


def short(name):
    for k in ("k_lm_step_quad", "k_lin_laser", "k_lin_imu", "k_lin_small", "k_compact_active", "k_lm_step_tw", "k_lm_step", "k_marg_schur",
              "k_lm_finish", "k_lm_begin", "k_group_offsets"):
        if k in name:
            return k
    return name[:24]


def main(db_path, first=60, count=6):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s
                              on d.kernel_id = s.id order by d.start"""))
    ev = [(short(n), a, b) for n, a, b in rows]
    quad = [i for i, e in enumerate(ev) if e[0] == "k_lm_step_quad"]
    print("kernels %d, quad launches %d" % (len(ev), len(quad)))
    gaps = []
    for qi in range(len(quad) - 1):
        i0, i1 = quad[qi], quad[qi + 1]
        busy_end = max(e[2] for e in ev[i0:i1])
        gaps.append((ev[i1][1] - busy_end) / 1e3)
        if first <= qi < first + count:
            t0 = ev[i0][2]
            print("iteration %d: step %.0f us" % (qi, (ev[i0][2] - ev[i0][1]) / 1e3))
            for e in ev[i0 + 1:i1]:
                print("   %-18s start %+8.1f  end %+8.1f  (%.1f us)" % (e[0], (e[1] - t0) / 1e3, (e[2] - t0) / 1e3, (e[2] - e[1]) / 1e3))
            print("   next step starts %+8.1f, idle before it %.1f us" % ((ev[i1][1] - t0) / 1e3, gaps[-1]))
    g = sorted(x for x in gaps if x < 1000.0)
    if g:
        print("idle before a step kernel: median %.1f us, mean %.1f us, max %.1f us over %d iterations" % (g[len(g) // 2], sum(g) / len(g), g[-1], len(g)))
    # union of busy time vs span over the longest run of consecutive quad launches
    if len(quad) > 2:
        a, b = ev[quad[0]][1], max(e[2] for e in ev[quad[0]:quad[-1]])
        segs = sorted((e[1], e[2]) for e in ev[quad[0]:quad[-1]])
        busy, cur_a, cur_b = 0, segs[0][0], segs[0][1]
        for s, t in segs[1:]:
            if s > cur_b:
                busy += cur_b - cur_a
                cur_a, cur_b = s, t
            else:
                cur_b = max(cur_b, t)
        busy += cur_b - cur_a
        print("span %.2f ms, some kernel running %.2f ms (%.1f %%)" % ((b - a) / 1e6, busy / 1e6, 100.0 * busy / (b - a)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60, int(sys.argv[3]) if len(sys.argv) > 3 else 6)
