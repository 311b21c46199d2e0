#!/usr/bin/env python3
"""Reference tables at the sizes where the engine's default code path switches
(tests/golden/golden_large.json; run in the dev container, minutes to an hour).

The engine takes the MSD first sort from 2^25 entries, the pair path, the
windowed rank table and the doubling rounds by what the input holds -- none of
which the reference's own small fixtures reach by default.  This script feeds
seeded synthetic sequences (genometools_amd/synth.py: the very bytes the GPU
tests regenerate) as FASTA to oracle/_ref/gt_ref_sfx -- the reference's engine
compiled from /root/reference by oracle/Makefile.ref -- and stores md5 + size
of INDEX.suf/.lcp/.llv/.bwt and the text of INDEX.prj.  Only data is stored.

usage: make_golden_large.py [name ...]     (default: every case not yet stored)
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "oracle", "_ref", "gt_ref_sfx")
OUT = os.path.join(ROOT, "tests", "golden", "golden_large.json")

# name: (model, seed, n, protein)
CASES = {
    "uniform_64m": ("MODEL_UNIFORM_DNA", 42, 64 * 1000 * 1000, False),
    "humanlike_40m": ("MODEL_HUMANLIKE_DNA", 43, 40 * 1000 * 1000, False),
    "repeatheavy_40m": ("MODEL_REPEAT_HEAVY", 45, 40 * 1000 * 1000, False),
    "protein_64m": ("MODEL_PROTEIN", 44, 64 * 1000 * 1000, True),
    "humanlike_256m": ("MODEL_HUMANLIKE_DNA", 43, 256 * 1000 * 1000, False),
}


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    from genometools_amd import synth
    if not os.path.exists(BIN):
        sys.exit("build oracle/_ref first: make -f oracle/Makefile.ref")
    golden = json.load(open(OUT)) if os.path.exists(OUT) else {}
    names = sys.argv[1:] or [k for k in CASES if k not in golden]
    tmproot = "/dev/shm" if os.path.isdir("/dev/shm") else None
    for name in names:
        model, seed, n, protein = CASES[name]
        with tempfile.TemporaryDirectory(dir=tmproot) as tmp:
            t0 = time.time()
            enc = synth.generate(getattr(synth, model), seed, n)
            fasta = os.path.join(tmp, name + (".faa" if protein else ".fna"))
            synth.write_fasta(fasta, enc, protein=protein)
            del enc
            t1 = time.time()
            idx = os.path.join(tmp, "idx")
            subprocess.run([BIN, "-protein" if protein else "-dna", "-suf", "-lcp", "-bwt",
                            "-indexname", idx, "-db", fasta], check=True)
            t2 = time.time()
            entry = {"model": model, "seed": seed, "n": n,
                     "alphabet": "protein" if protein else "dna", "tables": {},
                     "reference_seconds": round(t2 - t1, 1)}
            for ext in ("suf", "lcp", "llv", "bwt"):
                p = idx + "." + ext
                entry["tables"][ext] = {"md5": md5(p), "bytes": os.path.getsize(p)}
            with open(idx + ".prj") as f:
                entry["prj"] = f.read()
            golden[name] = entry
            print("%s: generated in %.0f s, reference %.0f s" % (name, t1 - t0, t2 - t1), flush=True)
        with open(OUT, "w") as f:
            json.dump(golden, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
