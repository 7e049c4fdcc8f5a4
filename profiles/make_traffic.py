#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a
pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Units and corrections as that guide prescribes:
both counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes read (64 B tallied per
128-B request) -> doubled; WRITE_SIZE is taken as is (it matches known write volumes of the
streaming kernels here: normalise writes nnz*4 B and WRITE_SIZE reports exactly that).

Usage: make_traffic.py FETCH_results.db WRITE_results.db workload-tag [pipeline-steps-in-the-profiled-run | auto] > profiles/rNN_traffic_<tag>.json
(`auto`, the default: the launches of k_gene_moments — the moments pass runs exactly once per pipeline step; the r05 table was made with
a hand-counted 4 where the bench had run 7 steps)"""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda s: [x for x in t if s in x][0]
    kd, ks, pe, pi = T("kernel_dispatch"), T("kernel_symbol"), T("pmc_event"), T("info_pmc")
    rows = c.execute(f"""select s.display_name, sum(e.value), count(distinct d.id)
        from {pe} e join {pi} p on e.pmc_id = p.id join {kd} d on d.event_id = e.event_id
        join {ks} s on d.kernel_id = s.id where p.name = ? group by s.display_name""", (counter,)).fetchall()
    return {name: (val / n, n) for name, val, n in rows}


def main(fetch_db, write_db, tag, steps):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    if steps is None:
        once = [n for name, (_, n) in f.items() if "k_gene_moments" in name]
        steps = float(max(once)) if once else 1.0
    out = {"workload": tag, "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes); "
           "KiB -> bytes; FETCH x2 (gfx950 correction); per-launch averages", "kernels": {}}
    for name in sorted(set(f) | set(w)):
        fb = f.get(name, (0.0, 0))[0] * 1024.0 * 2.0
        wb = w.get(name, (0.0, 0))[0] * 1024.0
        n = f.get(name, w.get(name))[1]
        out["kernels"][name] = {"launches": n, "launches_per_step": n / steps, "fetch_bytes": fb, "write_bytes": wb,
                                "hbm_bytes": fb + wb}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], None if len(sys.argv) < 5 or sys.argv[4] == "auto" else float(sys.argv[4]))
