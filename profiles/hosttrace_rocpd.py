#!/usr/bin/env python3
"""Host API calls and kernel dispatches of the LAST pipeline step of a rocprofv3 rocpd database (--kernel-trace
--hip-runtime-trace), merged on one clock: where the host is when the GPU idles.
Usage: hosttrace_rocpd.py results.db [anchor-kernel-substring] [from_us] [to_us]"""
import sqlite3
import sys


def main(path, anchor="k_row_sum", lo=None, hi=None):
    c = sqlite3.connect(path)
    ker = c.execute("select s.display_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                    "on d.kernel_id = s.id order by d.start").fetchall()
    starts = [i for i, r in enumerate(ker) if anchor in r[0]]
    t0 = ker[starts[-1]][1]
    api = c.execute("select name, start, end, tid from regions order by start").fetchall()
    ev = []
    for n, s, e in ker:
        if s >= t0:
            ev.append((s, "GPU ", n.split("(")[0].replace("void ", "").replace("srx::", "")[:44], e - s))
    # the host calls that queued this step start before its first kernel: one step's worth of lead
    for n, s, e, tid in api:
        if e >= t0 - 2_000_000:
            ev.append((s, "host", n, e - s))
    ev.sort()
    lo = float(lo) if lo is not None else -1e9
    hi = float(hi) if hi is not None else 1e9
    print("| at us | where | what | dur us |")
    print("|---|---|---|---|")
    for s, w, n, d in ev:
        at = (s - t0) / 1e3
        if lo <= at <= hi:
            print(f"| {at:.1f} | {w} | `{n}` | {d / 1e3:.1f} |")


if __name__ == "__main__":
    main(*sys.argv[1:])
