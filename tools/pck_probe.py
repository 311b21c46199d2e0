#!/usr/bin/env python3
"""Time the packed-index construction (INDEX.bdx image on the device) behind an
ESA build of a synthetic sequence.

  python tools/pck_probe.py --symbols 3e9 [--model 1] [--locbitmap yes|no] [--reps 3]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from genometools_amd import _lib, esa, pck, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--symbols", type=float, default=1e9)
    ap.add_argument("--model", type=int, default=synth.MODEL_HUMANLIKE_DNA)
    ap.add_argument("--seed", type=int, default=43)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--bsize", type=int, default=8)
    ap.add_argument("--blbuck", type=int, default=8)
    ap.add_argument("--locfreq", type=int, default=16)
    ap.add_argument("--locbitmap", default=None)
    a = ap.parse_args()
    n = int(a.symbols)
    lib = _lib.load()
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    _lib.check(lib.gtamd_synth_bytes(0, a.model, a.seed, n, buf.data_ptr()))
    eng = esa.EsaEngine(n, synth.numofchars(a.model))
    eng.set_sequence_device(buf.data_ptr(), n)
    del buf
    torch.cuda.empty_cache()
    eng.run(esa.WANT_SUF | esa.WANT_BWT)
    bm = None if a.locbitmap is None else a.locbitmap == "yes"
    with pck.PackedIndex() as p:
        out = []
        for _ in range(a.reps):
            p.build_from_esa(eng, bsize=a.bsize, blbuck=a.blbuck, locfreq=a.locfreq, locbitmap=bm)
            out.append(p.info())
        inf = out[-1]
        ms = min(o["build_ms"] for o in out)
        # algorithmic bytes: BWT read twice (1 B), suffix array read twice (8 B)
        # when locate information is stored, the image written once
        alg = n * (2 * 1 + (2 * 8 if a.locfreq else 0)) + inf["file_bytes"]
        print(json.dumps({"n": n, "model": a.model, "options": vars(a), "info": inf,
                          "build_ms_all": [o["build_ms"] for o in out], "build_ms": ms,
                          "gbp_per_s": n / ms / 1e6, "algorithmic_GBps": alg / ms / 1e6,
                          "esa_ms": eng.timing()["total_ms"]}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
