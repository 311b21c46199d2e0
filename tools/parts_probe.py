#!/usr/bin/env python3
"""Dev tool: run a sharded build with R engine contexts on ONE device (thread
transport of tests/thread_comm.py) and print the per-part stage times.  The
parts share the device, so the absolute times are pessimistic; what the probe
shows is how the work of a part changes with R."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from genometools_amd import _lib, synth  # noqa: E402
import thread_comm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=float, default=64e6)
ap.add_argument("--model", type=int, default=1)
ap.add_argument("--parts", default="2")
ap.add_argument("--want", type=int, default=7)
ap.add_argument("--serial", action="store_true",
                help="one part computes at a time: the time a part holds the device is "
                     "what it would need on a GPU of its own")
a = ap.parse_args()
n = int(a.n)
lib = _lib.load()
buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
_lib.check(lib.gtamd_synth_bytes(0, a.model, 43, n, buf.data_ptr()))
enc = buf.cpu().numpy()
del buf
torch.cuda.empty_cache()
N = n + 1
for parts in [int(x) for x in a.parts.split(",")]:
    # two sequences through the same contexts: the second run is warm
    views = [None] * parts
    out = thread_comm.build_sequences_in_parts([enc, enc], synth.numofchars(a.model), parts, a.want,
                                               serial=a.serial, views=views)
    for rep, (tabs, stats, res) in enumerate(out):
        for r, (off, rr, compute_s) in enumerate(res):
            st, tm = rr.stats, rr.timing
            print("R=%d rep%d part%d slice %d tied %d pairs %d mem %.1f GB | %s total %.1f keygen+exchange %.1f sort %.1f "
                  "fin %.1f refine %.1f fix %.1f | sent %.2f GB in %d exchanges" % (
                      parts, rep, r, tm["scatter_items"], st["tied_suffixes"], st["pair_suffixes"],
                      st["device_bytes"] / 1e9,
                      ("OUTSIDE COLLECTIVES %.1f ms |" % (tm["total_ms"] - tm["comm_ms"])) if a.serial else "",
                      tm["total_ms"], tm["keygen_ms"], tm["sort_ms"],
                      tm["finalize_ms"], tm["refine_ms"], tm["tie_fix_ms"],
                      tm["comm_bytes"] / 1e9, tm["comm_calls"]), flush=True)
    if tabs["suf"] is not None:
        assert int(tabs["suf"].sum(dtype=np.uint64)) == (N * (N - 1) // 2) % (1 << 64)
        print("R=%d checksum ok" % parts, flush=True)
    del out, tabs
