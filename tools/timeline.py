#!/usr/bin/env python3
"""Timeline of the LAST engine build in a `rocprofv3 --kernel-trace` run (sqlite
.db of this image): one line per kernel launch -- start since the build's first
kernel, duration, idle gap before it (all streams merged), name.  Dev tool:

  python tools/timeline.py RESULTS.db [--first k_msd_hist_a] > gpurun_out/timeline.txt
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--first", default="k_msd_hist_a",
                    help="kernel that opens a build (its last launch starts the timeline)")
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, start, end from kernels order by start" % name).fetchall()
    t0 = None
    for n, s, e in rows:
        if n.startswith(a.first) or (" " + a.first) in n:
            t0 = s
    if t0 is None:
        t0 = rows[0][1]
    busy_until = t0
    for n, s, e in rows:
        if s < t0:
            continue
        gap = max(0, s - busy_until)
        busy_until = max(busy_until, e)
        short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        print("%9.3f %8.3f %7.3f %s" % ((s - t0) / 1e6, (e - s) / 1e6, gap / 1e6, short))


if __name__ == "__main__":
    main()
