#!/usr/bin/env python3
"""Per-kernel PMC sums from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --pmc ... -o NAME`).
Usage: python profiles/summarize_pmc.py gpurun_out/pmc/x_results.db [kernel-substring]"""
import collections
import sqlite3
import sys


def main(path, filt=""):
    c = sqlite3.connect(path)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda s: [x for x in t if s in x][0]
    kd, ks, pe, pi = T("kernel_dispatch"), T("kernel_symbol"), T("pmc_event"), T("info_pmc")
    rows = c.execute(f"""select s.display_name, p.name, sum(e.value), count(distinct d.id), sum(d.end - d.start)
        from {pe} e join {pi} p on e.pmc_id = p.id join {kd} d on d.event_id = e.event_id
        join {ks} s on d.kernel_id = s.id group by s.display_name, p.name""").fetchall()
    per = collections.defaultdict(dict)
    meta = {}
    for name, pmc, val, n, dur in rows:
        per[name][pmc] = val
        meta[name] = (n, dur)
    pmcs = sorted({p for v in per.values() for p in v})
    print("| kernel | calls | " + " | ".join(pmcs) + " |")
    print("|---|---|" + "---|" * len(pmcs))
    for name in sorted(per, key=lambda k: -max(per[k].values())):
        if filt and filt not in name:
            continue
        n, _ = meta[name]
        short = name.replace("|", "\\|")
        short = short[:70] + ("..." if len(short) > 70 else "")
        print(f"| `{short}` | {n} | " + " | ".join(f"{per[name].get(p, 0) / n:.4g}" for p in pmcs) + " |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
