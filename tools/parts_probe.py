#!/usr/bin/env python3
"""Dev tool: run a sharded build with R engine contexts on ONE device (thread
transport of tests/thread_comm.py) and print the per-part stage times."""
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
ap.add_argument("--parts", type=int, default=2)
a = ap.parse_args()
n = int(a.n)
lib = _lib.load()
buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
_lib.check(lib.gtamd_synth_bytes(0, a.model, 43, n, buf.data_ptr()))
enc = buf.cpu().numpy()
del buf
for rep in range(2):
    tabs, stats, per = thread_comm.build_in_parts(enc, synth.numofchars(a.model), a.parts, timing=True)
    for r, (st, tm) in enumerate(per):
        print("rep%d part%d entries-tied %d | total %.1f keygen %.1f sort %.1f fin %.1f refine %.1f fix %.1f" % (
            rep, r, st["tied_suffixes"], tm["total_ms"], tm["keygen_ms"], tm["sort_ms"],
            tm["finalize_ms"], tm["refine_ms"], tm["tie_fix_ms"]))
N = n + 1
assert int(tabs["suf"].sum()) == N * (N - 1) // 2
print("checksum ok")
