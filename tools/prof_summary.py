#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / min / max, like --stats CSV.
usage: python tools/prof_summary.py gpurun_out/prof/x_results.db [--md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                      "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    md = "--md" in sys.argv
    if md:
        print("| kernel | calls | total us | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
    for r in rows:
        name = r[0].replace("void ynk::", "").replace("ynk::", "")
        name = name.split("(")[0] if "(" in name and "<" not in name.split("(")[0][-1:] else name[:64]
        if md:
            print("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (name[:64], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
        else:
            print("%-64s n=%6d total_us=%10.1f avg_us=%8.2f min=%8.2f max=%8.2f %5.1f%%" % (name[:64], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))


if __name__ == "__main__":
    main()
