#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace results database (rocpd sqlite) as a per-kernel stats CSV
(name, calls, total_ms, avg_us, min_us, max_us, pct) — the same columns `--stats` prints."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    q = """select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc"""
    rows = list(db.execute(q))
    tot = sum(r[2] for r in rows) or 1.0
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,pct"]
    for r in rows:
        lines.append('"%s",%d,%.3f,%.3f,%.3f,%.3f,%.2f' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
