#!/usr/bin/env python3
"""Dispatch timeline of the LAST pipeline step in a rocprofv3 rocpd database: kernel, start offset, duration
and the idle gap before it.  Usage: timeline_rocpd.py results.db [anchor-kernel-substring]"""
import sqlite3
import sys


def main(path, anchor="k_row_pass"):
    c = sqlite3.connect(path)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda s: [x for x in t if s in x][0]
    kd, ks = T("kernel_dispatch"), T("kernel_symbol")
    rows = c.execute(f"select s.display_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id "
                     f"order by d.start").fetchall()
    starts = [i for i, r in enumerate(rows) if anchor in r[0]]
    rows = rows[starts[-1]:]
    t0 = rows[0][1]
    prev_end = t0
    busy = gaps = 0.0
    print("| # | kernel | start us | dur us | gap before us |")
    print("|---|---|---|---|---|")
    for i, (name, s, e) in enumerate(rows):
        short = name.split("(")[0].replace("void ", "").replace("srx::", "")[:40]
        gap = (s - prev_end) / 1e3
        print(f"| {i} | `{short}` | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap:.1f} |")
        busy += (e - s) / 1e3
        gaps += max(gap, 0.0)
        prev_end = max(prev_end, e)
    print(f"\nspan {(prev_end - t0) / 1e3:.1f} us, kernels {busy:.1f} us, idle gaps {gaps:.1f} us, {len(rows)} dispatches")


if __name__ == "__main__":
    main(*sys.argv[1:])
