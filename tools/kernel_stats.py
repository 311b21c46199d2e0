#!/usr/bin/env python3
"""Per-kernel summary of a `rocprofv3 --kernel-trace --stats` run (this image
writes a sqlite .db, not CSV files).

  python tools/kernel_stats.py RESULTS.db --builds 4 > profiles/rNN_..._kernel_stats.csv

--builds: engine builds inside the profiled command (warmup + steps), so that
per_build_ms is comparable with bench.py's ms_per_step."""
import argparse
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--builds", type=int, default=1)
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(
        "select %s, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
        "from kernels group by %s order by 3 desc" % (name, name)).fetchall()
    out = sys.stdout
    out.write("kernel,calls,total_ms,per_build_ms,avg_ms,min_ms,max_ms\n")
    tot = 0.0
    for n, c, s, av, mn, mx in rows:
        tot += s
        out.write('"%s",%d,%.3f,%.3f,%.4f,%.4f,%.4f\n'
                  % (n, c, s / 1e6, s / 1e6 / a.builds, av / 1e6, mn / 1e6, mx / 1e6))
    out.write('"TOTAL",,%.3f,%.3f,,,\n' % (tot / 1e6, tot / 1e6 / a.builds))


if __name__ == "__main__":
    main()
