#!/usr/bin/env python3
"""Time the engine on a text of low-copy repeats only: uniform DNA in which every
second block of 8192 symbols is a copy of the block before it -- half of the
suffixes are pairs, almost nothing is left for the doubling rounds.  Dev tool
(the case in which the pairs' table entries have no rounds to hide behind)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometools_amd import _lib, esa, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=float, default=1e9)
    ap.add_argument("--runs", type=int, default=6)
    ap.add_argument("--ab", default="GTAMD_APPLY_EARLY")
    ap.add_argument("--abvals", default="0,2")
    a = ap.parse_args()
    n = int(a.n) // 16384 * 16384
    lib = _lib.load()
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    _lib.check(lib.gtamd_synth_bytes(0, synth.MODEL_UNIFORM_DNA, 5, n, buf.data_ptr()))
    v = buf.view(-1, 2, 8192)
    v[:, 1, :] = v[:, 0, :]
    torch.cuda.synchronize()
    eng = esa.EsaEngine(n, 4)
    eng.set_sequence_device(buf.data_ptr(), n)
    vals = a.abvals.split(",")
    for r in range(a.runs):
        os.environ[a.ab] = vals[r % len(vals)]
        eng.run(7)
        tm, st = eng.timing(), eng.stats()
        print("[%s=%s] dev %.1f ms | sort %.1f refine %.1f fix %.1f | tied %d (pairs %d) rounds %d" % (
            a.ab, os.environ[a.ab], tm["total_ms"], tm["sort_ms"], tm["refine_ms"], tm["tie_fix_ms"],
            st["tied_suffixes"], st["pair_suffixes"], st["refine_rounds"]), flush=True)


if __name__ == "__main__":
    main()
