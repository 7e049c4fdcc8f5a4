#!/usr/bin/env python3
"""Kernel-stats table (calls / total / avg / min / max, VGPRs, LDS) from a rocprofv3 rocpd
database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes NAME_results.db).
Usage: python profiles/summarize_rocpd.py gpurun_out/prof/x_results.db > profiles/rNN_x.md"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if "kernel_dispatch" in t][0]
    ks = [t for t in tables if "kernel_symbol" in t][0]
    rows = c.execute(f"""
        select s.display_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start),
               max(d.end - d.start), s.arch_vgpr_count, s.sgpr_count, max(d.group_segment_size),
               max(d.private_segment_size), max(d.grid_size_x), max(d.workgroup_size_x)
        from {kd} d join {ks} s on d.kernel_id = s.id group by s.display_name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | scratch B | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = r[0].replace("|", "\\|")
        if len(name) > 90:
            name = name[:87] + "..."
        print(f"| `{name}` | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.1f} | {r[4] / 1e3:.1f} | {r[5] / 1e3:.1f} | "
              f"{100 * r[2] / total:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |")


if __name__ == "__main__":
    main(sys.argv[1])
