#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc runs (sqlite output) per kernel and write
profiles/traffic.json for bench.py.

  python tools/pmc_summary.py FETCH.db WRITE.db --n 3000000000 --model 1 \
      --md profiles/r02_pmc_hbm_bytes.md --json profiles/traffic.json

FETCH.db / WRITE.db: results of two separate passes (`rocprofv3 --kernel-trace
--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; the two counters do not fit one pass,
MI355X_MICROARCH.md).  Counter unit: 1024 B... as reported by rocprofv3 (KB);
FETCH_SIZE is doubled for the wide streaming reads of the sort kernels, as the
guide prescribes for gfx950."""
import argparse
import hashlib
import json
import os
import sqlite3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(db, counter):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
    # rocprofv3's view: one row per (dispatch, counter)
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = cur.execute(
        "select dispatch_id, %s, counter_name, sum(value) from counters_collection "
        "where counter_name = ? group by dispatch_id, %s, counter_name order by dispatch_id"
        % (name_col, name_col), (counter,)).fetchall()
    return [(r[0], r[1], r[3]) for r in rows]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_db")
    ap.add_argument("write_db")
    ap.add_argument("--n", type=int, required=True)
    ap.add_argument("--model", type=int, required=True)
    ap.add_argument("--md")
    ap.add_argument("--json")
    ap.add_argument("--unit", type=float, default=1024.0, help="bytes per counter unit")
    ap.add_argument("--bench-json", help="a bench.py line of the same workload: its roofline.entries_read_per_launch / "
                                         "entries_written_per_launch give the algorithmic bytes of k_msd_local")
    a = ap.parse_args()
    fetch = per_dispatch(a.fetch_db, "FETCH_SIZE")
    write = per_dispatch(a.write_db, "WRITE_SIZE")
    agg = {}
    for kind, rows in (("f", fetch), ("w", write)):
        for _, name, v in rows:
            e = agg.setdefault(short(name), {"f": [], "w": []})
            e[kind].append(v * a.unit)
    lines = ["| kernel | launches | FETCH_SIZE GB raw (sum / largest) | WRITE_SIZE GB (sum / largest) |",
             "|---|---|---|---|"]
    for name, e in sorted(agg.items(), key=lambda kv: -(sum(kv[1]["f"]) + sum(kv[1]["w"]))):
        f, w = e["f"] or [0], e["w"] or [0]
        lines.append("| %s | %d | %.3f / %.3f | %.3f / %.3f |" % (
            name, max(len(e["f"]), len(e["w"])), sum(f) / 1e9, max(f) / 1e9, sum(w) / 1e9, max(w) / 1e9))
    text = "\n".join(lines)
    print(text)
    if a.md:
        with open(a.md, "a") as out:
            out.write(text + "\n")
    # the dominant kernel of the first sort (bench.py's roofline kernel): k_msd_local
    # in a DNA whole-table build (one launch per build), else the full passes of
    # k_rs_scatter<unsigned long, unsigned int, 1> (the largest launches)
    def sha(name):
        with open(os.path.join(ROOT, "genometools_amd", "csrc", name), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()
    note = ("separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE); FETCH_SIZE doubled per "
            "MI355X_MICROARCH.md; bench.py reports it only while %s has this hash")
    key = [k for k in agg if k.startswith("k_msd_local") and not k.startswith("k_msd_local_radix")]
    if a.json and key:
        e = agg[key[0]]
        fb = 2.0 * sum(e["f"]) / len(e["f"])
        wb = sum(e["w"]) / len(e["w"])
        # what the kernel reads (every run that fits its tile) and writes (the runs it
        # sorts itself), as the engine counted them: 8 B read, 14.125 B written per entry
        alg, alg_note = 22.125 * (a.n + 1), "22.125 B x N (no bench line given: upper bound)"
        if a.bench_json:
            with open(a.bench_json) as f:
                roof = json.loads([l for l in f if l.startswith("{")][-1])["roofline"]
            alg = 8.0 * roof["entries_read_per_launch"] + 14.125 * roof["entries_written_per_launch"]
            alg_note = "8 B x %.0f entries read + 14.125 B x %.0f entries written (bench.py, same workload)" % (
                roof["entries_read_per_launch"], roof["entries_written_per_launch"])
        with open(a.json, "w") as out:
            json.dump({"n": a.n, "model": a.model, "kernel": "k_msd_local",
                       "launch": "the one launch of a build, ~N=%d entries: mean over the builds profiled" % (a.n + 1),
                       "fetch_bytes_corrected": fb, "write_bytes": wb,
                       "hbm_bytes_per_launch": fb + wb,
                       "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_how": alg_note,
                       "kernel_source": "esa_msd.h", "kernel_source_sha256": sha("esa_msd.h"),
                       "note": note % "genometools_amd/csrc/esa_msd.h"}, out, indent=1)
            out.write("\n")
        return
    key = [k for k in agg if k.startswith("k_rs_scatter<unsigned long, unsigned int")]
    if a.json and key:
        e = agg[key[0]]
        big_f = sorted(e["f"])[-5:]
        big_w = sorted(e["w"])[-5:]
        fb = 2.0 * sum(big_f) / len(big_f)
        wb = sum(big_w) / len(big_w)
        with open(a.json, "w") as out:
            json.dump({"n": a.n, "model": a.model, "kernel": "k_rs_scatter",
                       "launch": "first-sort pass (full 8-bit digit), N=%d pairs: mean of the 5 largest launches" % (a.n + 1),
                       "fetch_bytes_corrected": fb, "write_bytes": wb,
                       "hbm_bytes_per_launch": fb + wb,
                       "algorithmic_bytes_per_launch": 24 * (a.n + 1),
                       "kernel_source": "esa_prims.hip", "kernel_source_sha256": sha("esa_prims.hip"),
                       "note": note % "genometools_amd/csrc/esa_prims.hip"}, out, indent=1)
            out.write("\n")


if __name__ == "__main__":
    main()
