#!/usr/bin/env python3
"""End-to-end timing of gt-suffixerator-amd on a synthetic FASTA file (GPU box):
FASTA on disk -> all index files, with the device reader and with the host
reader; checks that both write the same files.

  python tools/e2e_probe.py --symbols 1000000000 [--dir /dev/shm/e2e]
"""
import argparse
import hashlib
import os
import shutil
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from genometools_amd import _lib, synth  # noqa: E402


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--symbols", type=int, default=256_000_000)
    ap.add_argument("--dir", default="/dev/shm/gtamd_e2e")
    ap.add_argument("--encoders", default="device,host")
    ap.add_argument("--tables", default="-suf -lcp -bwt")
    ap.add_argument("--packedindex", action="store_true",
                    help="also time `packedindex mkindex` (FASTA -> INDEX.bdx) on the same file")
    ap.add_argument("--fastq", type=int, default=0, metavar="READLEN",
                    help="write the symbols as four-line FASTQ records of READLEN symbols instead "
                         "of FASTA (the device FASTQ reader against the host reader)")
    ap.add_argument("--generate-only", action="store_true",
                    help="write DIR/genome.fna and stop (input for a profiler run)")
    a = ap.parse_args()
    lib = _lib.load()
    os.makedirs(a.dir, exist_ok=True)
    try:
        n = a.symbols
        buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
        _lib.check(lib.gtamd_synth_bytes(0, synth.MODEL_HUMANLIKE_DNA, 43, n, buf.data_ptr()))
        enc = buf.cpu().numpy()
        del buf
        torch.cuda.empty_cache()
        fa = os.path.join(a.dir, "genome.fna")
        t = time.time()
        if a.fastq:
            # reads of READLEN symbols cut from the sequence (separators and wildcards
            # become N), a name and a quality line each; written in slabs
            L = a.fastq
            lut = np.frombuffer(b"ACGT" + b"N" * 252, dtype=np.uint8)
            nrec = n // L
            name_w = 12
            rec = 1 + name_w + 1 + L + 1 + 1 + 1 + L + 1          # @name\n seq\n +\n qual\n
            with open(fa, "wb") as f:
                for r0 in range(0, nrec, 1 << 18):
                    r1 = min(nrec, r0 + (1 << 18))
                    k = r1 - r0
                    slab = np.empty((k, rec), dtype=np.uint8)
                    slab[:, 0] = ord("@")
                    names = np.char.zfill(np.arange(r0, r1).astype("U"), name_w - 1).astype("S")
                    slab[:, 1] = ord("r")
                    slab[:, 2:1 + name_w] = np.frombuffer(b"".join(names.tolist()), dtype=np.uint8).reshape(k, name_w - 1)
                    slab[:, 1 + name_w] = 10
                    slab[:, 2 + name_w:2 + name_w + L] = lut[enc[r0 * L:r1 * L]].reshape(k, L)
                    slab[:, 2 + name_w + L] = 10
                    slab[:, 3 + name_w + L] = ord("+")
                    slab[:, 4 + name_w + L] = 10
                    slab[:, 5 + name_w + L:5 + name_w + 2 * L] = 33 + (np.arange(k * L, dtype=np.uint32).reshape(k, L) * 2654435761 >> 26 & 63).astype(np.uint8)
                    slab[:, 5 + name_w + 2 * L] = 10
                    f.write(slab.tobytes())
            n = nrec * L
        else:
            synth.write_fasta(fa, enc)
        del enc
        print("%s: %d symbols, %.1f MB, written in %.1f s" % ("FASTQ" if a.fastq else "FASTA", n, os.path.getsize(fa) / 1e6,
                                                                time.time() - t), flush=True)
        if a.generate_only:
            a.dir = None
            return 0
        cli = os.path.join(_lib.HERE, "gt-suffixerator-amd")
        sums = {}
        for encoder in a.encoders.split(","):
            idx = os.path.join(a.dir, "idx_" + encoder)
            t = time.time()
            r = subprocess.run([cli, "-dna", *a.tables.split(), "-v", "-encoder", encoder,
                                "-indexname", idx, "-db", "genome.fna"], cwd=a.dir,
                               capture_output=True, text=True)
            wall = time.time() - t
            if r.returncode != 0:
                print(encoder, "FAILED", r.stderr)
                return 1
            line = [l for l in r.stdout.splitlines()
                    if l.startswith(("# seconds", "# device encoder", "# device reader"))]
            print("%s reader: wall %.2f s\n  %s" % (encoder, wall, "\n  ".join(line)),
                  flush=True)
            sums[encoder] = {ext: md5(idx + "." + ext) for ext in
                             ("esq", "ssp", "des", "sds", "md5", "prj", "suf", "lcp", "bwt")
                             if os.path.exists(idx + "." + ext)}
            for ext in ("suf", "lcp", "llv", "bwt", "esq"):
                if os.path.exists(idx + "." + ext):
                    os.unlink(idx + "." + ext)
        if a.packedindex:
            idx = os.path.join(a.dir, "pck")
            t = time.time()
            r = subprocess.run([cli, "packedindex", "mkindex", "-dna", "-v", "-indexname", idx,
                                "-db", "genome.fna"], cwd=a.dir, capture_output=True, text=True)
            wall = time.time() - t
            if r.returncode != 0:
                print("packedindex mkindex FAILED", r.stderr)
                return 1
            line = [l for l in r.stdout.splitlines() if l.startswith(("# seconds", "# packed index"))]
            print("packedindex mkindex: wall %.2f s, INDEX.bdx %.1f MB\n  %s" % (
                wall, os.path.getsize(idx + ".bdx") / 1e6, "\n  ".join(line)), flush=True)
        if len(sums) == 2:
            same = sums["device"] == sums["host"]
            print("device and host reader wrote identical files:", same)
            if not same:
                return 1
    finally:
        if a.dir is not None:
            shutil.rmtree(a.dir, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
