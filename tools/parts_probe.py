#!/usr/bin/env python3
"""Dev tool: run a sharded build with R engine contexts on ONE device (thread
transport of tests/thread_comm.py) and print, per part, the stage times, the
bytes it sent and the number of exchanges it took part in.  With --serial one
part computes at a time (the parts hand the device over at every collective), so
"outside collectives" is what the part would need on a GPU of its own.  The
tables stay on the device (3 Gbp: 30 GB); they are checked there: the slices
tile the table and the suffix tables add up to N (N - 1) / 2."""
import argparse
import json
import os
import sys
import threading

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from genometools_amd import _lib, esa, synth  # noqa: E402
from genometools_amd.dist import _DevMem  # noqa: E402
import thread_comm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=float, default=64e6)
ap.add_argument("--model", type=int, default=1)
ap.add_argument("--seed", type=int, default=43)
ap.add_argument("--parts", default="2")
ap.add_argument("--want", type=int, default=7)
ap.add_argument("--reps", type=int, default=2, help="runs per part count (the first is cold)")
ap.add_argument("--serial", action="store_true")
ap.add_argument("--json", help="append one JSON record per part count to this file")
a = ap.parse_args()
n = int(a.n)
N = n + 1
sigma = synth.numofchars(a.model)
lib = _lib.load()
dev = torch.device("cuda", 0)
buf = torch.empty(n, dtype=torch.uint8, device=dev)
_lib.check(lib.gtamd_synth_bytes(0, a.model, a.seed, n, buf.data_ptr()))
torch.cuda.synchronize()

for parts in [int(x) for x in a.parts.split(",")]:
    shared = thread_comm.ThreadComm(parts, 0, a.serial)
    rows = [[None] * parts for _ in range(a.reps)]
    errors = []

    def worker(r):
        try:
            with esa.EsaEngine(n, sigma, 0) as eng:
                view = shared.view(r)
                eng.set_sequence_device(buf.data_ptr(), n)
                for rep in range(a.reps):
                    eng.set_part(r, parts, view)
                    view.compute_s, view.bytes_exchanged, view.calls = 0.0, 0, 0
                    view.begin()
                    try:
                        eng.run(a.want)
                    finally:
                        view.end()
                    st, tm = eng.stats(), eng.timing()
                    ent = eng.entries(esa.TAB_SUF)
                    s = 0
                    if a.want & 1 and ent:
                        t = torch.as_tensor(_DevMem(eng.device_pointer(esa.TAB_SUF), ent * 8), device=dev)
                        s = int(t.view(torch.int64).sum().item()) & ((1 << 64) - 1)
                    rows[rep][r] = dict(offset=eng.table_offset(), entries=ent, sufsum=s, stats=st, timing=tm,
                                        compute_ms=view.compute_s * 1e3)
        except Exception as e:   # noqa: BLE001
            errors.append((r, repr(e)))
            shared.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(parts)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for rep in range(a.reps):
        expect, total = 0, 0
        for r, row in enumerate(rows[rep]):
            st, tm = row["stats"], row["timing"]
            assert row["offset"] == expect, "slices must tile the table"
            expect += row["entries"]
            total = (total + row["sufsum"]) & ((1 << 64) - 1)
            outside = tm["total_ms"] - tm["comm_ms"]
            print("R=%d rep%d part%d slice %d tied %d pairs %d rounds %d mem %.1f GB | %stotal %.1f first sort %.1f "
                  "(level A / filter %.1f) refine %.1f fix %.1f | sent %.3f GB in %d exchanges" % (
                      parts, rep, r, row["entries"], st["tied_suffixes"], st["pair_suffixes"], st["refine_rounds"],
                      st["device_bytes"] / 1e9,
                      ("OUTSIDE COLLECTIVES %.1f ms (held the device %.1f ms) | " % (outside, row["compute_ms"]))
                      if a.serial else "",
                      tm["total_ms"], tm["keygen_ms"] + tm["sort_ms"] + tm["finalize_ms"], tm["keygen_ms"],
                      tm["refine_ms"], tm["tie_fix_ms"], tm["comm_bytes"] / 1e9, tm["comm_calls"]), flush=True)
        assert expect == N
        if a.want & 1:
            assert total == (N * (N - 1) // 2) % (1 << 64), "suffix tables do not add up"
        print("R=%d rep%d: slices tile the table, checksum ok" % (parts, rep), flush=True)
    if a.json:
        last = rows[-1]
        rec = {"n": n, "model": a.model, "seed": a.seed, "parts": parts, "serial": a.serial,
               "outside_collectives_ms": [x["timing"]["total_ms"] - x["timing"]["comm_ms"] for x in last],
               "held_device_ms": [x["compute_ms"] for x in last],
               "sent_bytes": [x["timing"]["comm_bytes"] for x in last],
               "exchanges": [x["timing"]["comm_calls"] for x in last],
               "slice_entries": [x["entries"] for x in last],
               "refine_rounds": max(x["stats"]["refine_rounds"] for x in last)}
        with open(a.json, "a") as f:
            f.write(json.dumps(rec) + "\n")
    del rows
    torch.cuda.empty_cache()
