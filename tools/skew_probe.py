#!/usr/bin/env python3
"""Build time on a sequence with a skewed base composition (real genomes are
AT-rich: the 8-mer ranges after level B of the first sort differ in size by large
factors), for several depths of level C in one process.  Dev tool.

  python tools/skew_probe.py --n 3e9 --at 0.6 [--cbits 8,6,adaptive]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometools_amd import esa  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=float, default=1e9)
    ap.add_argument("--at", type=float, default=0.6, help="share of A + T")
    ap.add_argument("--cbits", default="8,6")
    ap.add_argument("--runs", type=int, default=2)
    a = ap.parse_args()
    n = int(a.n)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    # skewed letters on the device: u < at/2 -> A, < at -> T, then C / G halves
    step = 1 << 28
    g = torch.Generator(device="cuda:0")
    g.manual_seed(7)
    for o in range(0, n, step):
        m = min(step, n - o)
        u = torch.rand(m, device="cuda:0", generator=g)
        x = torch.full((m,), 2, dtype=torch.uint8, device="cuda:0")     # g
        x[u < a.at + (1 - a.at) / 2] = 1                                 # c
        x[u < a.at] = 3                                                  # t
        x[u < a.at / 2] = 0                                              # a
        buf[o:o + m] = x
    eng = esa.EsaEngine(n, 4)
    eng.set_sequence_device(buf.data_ptr(), n)
    del buf
    for r in range(a.runs):
        for cb in a.cbits.split(","):
            if cb == "adaptive":
                os.environ.pop("GTAMD_MSD_CBITS", None)
            else:
                os.environ["GTAMD_MSD_CBITS"] = cb
            eng.run()
            tm = eng.timing()
            print("cbits %s: total %.1f ms (keygen %.1f sort %.1f refine %.1f fix %.1f)" % (
                cb, tm["total_ms"], tm["keygen_ms"], tm["sort_ms"], tm["refine_ms"],
                tm["tie_fix_ms"]), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
