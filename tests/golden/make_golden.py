#!/usr/bin/env python3
"""Generate tests/golden/ from the REFERENCE itself (run in the dev container).

For every suffixerator fixture named in the reference's own test suite
(testsuite/gt_suffixerator_include.rb:119-143 DNA FASTA files, :290-308
protein files) this runs oracle/_ref/gt_ref_sfx -- the reference's engine
compiled from /root/reference by oracle/Makefile.ref -- with
`-suf -lcp -bwt` and stores

  * tests/golden/fixtures/<name>           the input (data file of the
                                           reference's test suite, <= 120 KB)
  * tests/golden/tables/<name>.{suf,lcp,llv,bwt}.gz   for inputs <= 16 KB
  * tests/golden/golden.json               md5 + size of every table and the
                                           full .prj text, for all inputs

so that the CPU restatement (oracle/) and the HIP path can be checked against
the reference's real output on machines where /root/reference does not exist.
Only data is stored: inputs and expected outputs.
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("GT_REFERENCE", "/root/reference")
BIN = os.path.join(ROOT, "oracle", "_ref", "gt_ref_sfx")
OUT = os.path.join(ROOT, "tests", "golden")

# gzip-compressed input (read through zlib by name, src/core/file.c:42-53)
GZ = ["ebola-genomes.fna.gz"]
DNA = ["Arabidopsis-C99826.fna", "Atinsert.fna",
       "Atinsert_seqrange_13-17_rev.fna", "Atinsert_seqrange_3-7.fna",
       "Atinsert_single_3.fna", "Atinsert_single_3_rev.fna",
       "Copysorttest.fna", "Duplicate.fna", "Ecoli-section1.fna",
       "Ecoli-section2.fna", "Random-Small.fna", "Random.fna",
       "Random159.fna", "Random160.fna", "RandomN.fna", "Reads1.fna",
       "Reads2.fna", "Reads3.fna", "Repfind-example.fna", "TTTN.fna",
       "Small.fna", "Smalldup.fna", "TTT-small.fna", "trna_glutamine.fna",
       "Verysmall.fna"]
PROTEIN = ["sw100K1.fsa", "sw100K2.fsa"]
# FASTQ inputs of the suite (testsuite/gt_suffixerator_include.rb:157-160)
FASTQ = ["fastq_long.fastq", "test10_multiline.fastq", "test1.fastq",
         "test5_tricky.fastq"]
# readmode / mirror variants (testsuite/gt_suffixerator_include.rb:17-56 -dir,
# :464-487 -mirrored) on a few fixtures; keys "<name>|<dir>|<mirrored>"
VARIANT_FILES = ["Atinsert.fna", "Duplicate.fna", "RandomN.fna", "TTTN.fna",
                 "Small.fna", "Verysmall.fna", "Reads1.fna"]
VARIANTS = [("rev", False), ("cpl", False), ("rcl", False), ("fwd", True),
            ("rcl", True)]
MULTI = [["part1.fna", "part2.fna"], ["reads_a.fastq", "reads_b.fastq"],
         ["long_reads.fastq"], ["part2.fna", "part1.fna", "part2.fna"]]
ALL_DNA_SAT = ["direct", "bit", "uchar", "ushort", "uint32"]
FORCED_SAT = [("Duplicate.fna", ALL_DNA_SAT), ("TTTN.fna", ALL_DNA_SAT),
              ("Atinsert_seqrange_3-7.fna", ALL_DNA_SAT),
              ("extra/starts_ends_special.fna", ALL_DNA_SAT),
              ("extra/gt_in_line.fna", ALL_DNA_SAT + ["eqlen"]),
              ("Reads1.fna", ["eqlen"]),
              ("extra/protein_specials.faa", ["direct", "bytecompress"]),
              ("extra/protein_long_x.faa", ["direct", "bytecompress"])]
BCK_FILES = [("Atinsert.fna", [0, 1, 2, 3, 7]), ("Duplicate.fna", [0, 1, 2, 5]),
             ("RandomN.fna", [0, 3, 6]), ("TTTN.fna", [0, 1, 2, 3]),
             ("Verysmall.fna", [0, 2]), ("Reads1.fna", [0, 4]),
             ("extra/starts_ends_special.fna", [0, 1, 2, 3, 4]),
             ("extra/long_runs.fna", [0, 8]), ("extra/lowercase_iupac.fna", [0, 3]),
             ("extra/protein_specials.faa", [0, 1, 2, 3]),
             ("extra/protein_long_x.faa", [0, 2, 3]), ("sw100K1.fsa", [0, 2])]
CLIPDESC_FILES = ["Atinsert.fna", "Duplicate.fna", "extra/blanks.fna", "extra/crlf.fna",
                  "test10_multiline.fastq", "Reads1.fna"]
SMAP_CASES = [("trans_dna.map", "Atinsert.fna"), ("trans_dna.map", "Duplicate.fna"),
              ("trans_dna.map", "extra/lowercase_iupac.fna"),
              ("prot5.map", "sw100K1.fsa"), ("prot5.map", "extra/protein_specials.faa"),
              ("prot5.map", "extra/protein_long_x.faa")]
LOSSLESS_FILES = ["Atinsert.fna", "Duplicate.fna", "RandomN.fna", "TTTN.fna", "Reads1.fna",
                  "extra/lowercase_iupac.fna", "extra/lowercase_across_records.fna",
                  "extra/long_runs.fna", "extra/protein_specials.faa", "sw100K1.fsa",
                  "test10_multiline.fastq", "ebola-genomes.fna.gz"]
MAX_FIXTURE = 120 * 1024     # bigger inputs: md5 of tables only, no copy
MAX_TABLES = 16 * 1024       # store full tables only for small inputs


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


SEQFILES = ("des", "sds", "md5", "esq", "ssp")


def seqfiles(idx):
    """md5 of the sequence-side files the encoder wrote (INDEX.ssp only exists
    for more than one sequence and not for the equal-length access type)"""
    return {ext: {"md5": md5(idx + "." + ext),
                  "bytes": os.path.getsize(idx + "." + ext)}
            for ext in SEQFILES if os.path.exists(idx + "." + ext)}


def run_ref(flag, srcs, idx, extra=()):
    """INDEX.esq stores the -db arguments as typed: run from the input's
    directory with bare file names so the stored names do not depend on where
    the reference checkout lives"""
    subprocess.run([BIN, flag, "-suf", "-lcp", "-bwt", *extra, "-indexname", idx,
                    "-db"] + [os.path.basename(x) for x in srcs], check=True,
                   cwd=os.path.dirname(srcs[0]))


def main():
    if not os.path.exists(BIN):
        sys.exit("build oracle/_ref first: make -f oracle/Makefile.ref")
    os.makedirs(os.path.join(OUT, "fixtures"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "tables"), exist_ok=True)
    golden = {}
    for name, flag in [(f, "-dna") for f in DNA + FASTQ + GZ] + [(f, "-protein") for f in PROTEIN]:
        src = os.path.join(REF, "testdata", name)
        if not os.path.exists(src):
            print("missing", src)
            continue
        size = os.path.getsize(src)
        with tempfile.TemporaryDirectory() as tmp:
            idx = os.path.join(tmp, "idx")
            run_ref(flag, [src], idx)
            entry = {"alphabet": flag[1:], "input_bytes": size,
                     "input_md5": md5(src),
                     "fixture": size <= MAX_FIXTURE, "tables": {}}
            for ext in ("suf", "lcp", "llv", "bwt"):
                p = idx + "." + ext
                entry["tables"][ext] = {"md5": md5(p), "bytes": os.path.getsize(p)}
            with open(idx + ".prj") as f:
                entry["prj"] = f.read()
            # sequence-side files the encoder writes by default
            entry["seqfiles"] = seqfiles(idx)
            if size <= MAX_FIXTURE:
                shutil.copyfile(src, os.path.join(OUT, "fixtures", name))
            if size <= MAX_TABLES:
                for ext in ("suf", "lcp", "llv", "bwt"):
                    with open(idx + "." + ext, "rb") as fi, \
                         gzip.GzipFile(os.path.join(OUT, "tables", name + "." + ext + ".gz"),
                                       "wb", mtime=0) as fo:
                        fo.write(fi.read())
        golden[name] = entry
        print(name, entry["tables"]["suf"]["md5"])
    # our own edge-case inputs (tests/golden/extra/), reference output
    extra_dir = os.path.join(OUT, "extra")
    for name in sorted(f for f in os.listdir(extra_dir)
                       if f.endswith((".fna", ".faa", ".fna.gz", ".fna.bz2"))):
        src = os.path.join(extra_dir, name)
        flag = "-protein" if name.endswith(".faa") else "-dna"
        with tempfile.TemporaryDirectory() as tmp:
            idx = os.path.join(tmp, "idx")
            run_ref(flag, [src], idx)
            entry = {"alphabet": flag[1:], "input_bytes": os.path.getsize(src),
                     "input_md5": md5(src), "fixture": True, "extra": True, "tables": {}}
            for ext in ("suf", "lcp", "llv", "bwt"):
                pth = idx + "." + ext
                entry["tables"][ext] = {"md5": md5(pth), "bytes": os.path.getsize(pth)}
            with open(idx + ".prj") as f:
                entry["prj"] = f.read()
            entry["seqfiles"] = seqfiles(idx)
        golden["extra/" + name] = entry
    # several input files in one index (tests/golden/multi/): file length
    # table, separators between files, FASTQ buffer accounting
    multi = {}
    multi_dir = os.path.join(OUT, "multi")
    for files in MULTI:
        with tempfile.TemporaryDirectory() as tmp:
            idx = os.path.join(tmp, "idx")
            run_ref("-dna", [os.path.join(multi_dir, f) for f in files], idx)
            entry = {"files": files, "tables": {}}
            for ext in ("suf", "lcp", "llv", "bwt"):
                pth = idx + "." + ext
                entry["tables"][ext] = {"md5": md5(pth), "bytes": os.path.getsize(pth)}
            with open(idx + ".prj") as f:
                entry["prj"] = f.read()
            entry["seqfiles"] = seqfiles(idx)
        multi["+".join(files)] = entry
    with open(os.path.join(OUT, "golden_multi.json"), "w") as f:
        json.dump(multi, f, indent=1, sort_keys=True)
    # reference-written INDEX.esq/.ssp with every access type forced (-sat),
    # kept as files: input of the .esq reader's tests (tests/golden/esq/)
    esq_dir = os.path.join(OUT, "esq")
    os.makedirs(esq_dir, exist_ok=True)
    for name, sats in FORCED_SAT:
        src = (os.path.join(OUT, name) if name.startswith("extra/")
               else os.path.join(REF, "testdata", name))
        flag = "-protein" if name.endswith(".faa") else "-dna"
        for sat in sats:
            with tempfile.TemporaryDirectory() as tmp:
                idx = os.path.join(tmp, "idx")
                subprocess.run([BIN, flag, "-sat", sat, "-indexname", idx, "-db",
                                os.path.basename(src)], check=True,
                               cwd=os.path.dirname(src))
                for ext in ("esq", "ssp"):
                    if os.path.exists(idx + "." + ext):
                        shutil.copyfile(idx + "." + ext, os.path.join(
                            esq_dir, "%s.%s.%s" % (os.path.basename(name), sat, ext)))
    # bucket table (-bck) for several prefix lengths and the 32-bit suffix table
    # (-suftabuint); keys "<name>|<pl>" (pl 0: the recommended prefixlength)
    bck = {}
    for name, pls in BCK_FILES:
        src = (os.path.join(OUT, name) if name.startswith("extra/")
               else os.path.join(REF, "testdata", name))
        flag = "-protein" if name.endswith((".faa", ".fsa")) else "-dna"
        for pl in pls:
            with tempfile.TemporaryDirectory() as tmp:
                idx = os.path.join(tmp, "idx")
                run_ref(flag, [src], idx, extra=["-bck", "-suftabuint"] +
                        (["-pl", str(pl)] if pl else []))
                with open(idx + ".prj") as f:
                    prj = f.read()
                bck["%s|%d" % (name, pl)] = {
                    "bck": {"md5": md5(idx + ".bck"), "bytes": os.path.getsize(idx + ".bck")},
                    "suf32": {"md5": md5(idx + ".suf"), "bytes": os.path.getsize(idx + ".suf")},
                    "prj": prj}
    with open(os.path.join(OUT, "golden_bck.json"), "w") as f:
        json.dump(bck, f, indent=1, sort_keys=True)
    # -smap FILE: alphabets from a symbol map (tests/golden/extra/*.map)
    smap = {}
    for mapname, name in SMAP_CASES:
        src = (os.path.join(OUT, name) if name.startswith("extra/")
               else os.path.join(REF, "testdata", name))
        with tempfile.TemporaryDirectory() as tmp:
            idx = os.path.join(tmp, "idx")
            subprocess.run([BIN, "-smap", os.path.join(OUT, "extra", mapname), "-suf", "-lcp",
                            "-bwt", "-bck", "-indexname", idx, "-db", os.path.basename(src)],
                           check=True, cwd=os.path.dirname(src))
            entry = {"tables": {}}
            for ext in ("suf", "lcp", "llv", "bwt", "bck"):
                entry["tables"][ext] = {"md5": md5(idx + "." + ext),
                                        "bytes": os.path.getsize(idx + "." + ext)}
            with open(idx + ".prj") as f:
                entry["prj"] = f.read()
            entry["seqfiles"] = seqfiles(idx)
        smap["%s|%s" % (mapname, name)] = entry
    with open(os.path.join(OUT, "golden_smap.json"), "w") as f:
        json.dump(smap, f, indent=1, sort_keys=True)
    # -lossless: INDEX.ois, exception counts in INDEX.esq, MD5 over the originals
    lossless = {}
    for name in LOSSLESS_FILES:
        src = (os.path.join(OUT, name) if name.startswith("extra/")
               else os.path.join(REF, "testdata", name))
        flag = "-protein" if name.endswith((".faa", ".fsa")) else "-dna"
        with tempfile.TemporaryDirectory() as tmp:
            idx = os.path.join(tmp, "idx")
            subprocess.run([BIN, flag, "-lossless", "-indexname", idx, "-db",
                            os.path.basename(src)], check=True, cwd=os.path.dirname(src))
            lossless[name] = {ext: {"md5": md5(idx + "." + ext),
                                    "bytes": os.path.getsize(idx + "." + ext)}
                              for ext in ("ois", "esq", "md5", "des", "sds", "ssp")
                              if os.path.exists(idx + "." + ext)}
    with open(os.path.join(OUT, "golden_lossless.json"), "w") as f:
        json.dump(lossless, f, indent=1, sort_keys=True)
    # -clipdesc: descriptions cut at the first white space
    clip = {}
    for name in CLIPDESC_FILES:
        src = (os.path.join(OUT, name) if name.startswith("extra/")
               else os.path.join(REF, "testdata", name))
        with tempfile.TemporaryDirectory() as tmp:
            idx = os.path.join(tmp, "idx")
            subprocess.run([BIN, "-dna", "-clipdesc", "-indexname", idx, "-db",
                            os.path.basename(src)], check=True, cwd=os.path.dirname(src))
            clip[name] = {ext: {"md5": md5(idx + "." + ext),
                                "bytes": os.path.getsize(idx + "." + ext)}
                          for ext in ("des", "sds")}
    with open(os.path.join(OUT, "golden_clipdesc.json"), "w") as f:
        json.dump(clip, f, indent=1, sort_keys=True)
    variants = {}
    for name in VARIANT_FILES:
        src = os.path.join(REF, "testdata", name)
        for d, mir in VARIANTS:
            with tempfile.TemporaryDirectory() as tmp:
                idx = os.path.join(tmp, "idx")
                cmd = [BIN, "-dna", "-suf", "-lcp", "-bwt", "-dir", d, "-db", src,
                       "-indexname", idx] + (["-mirrored"] if mir else [])
                subprocess.run(cmd, check=True)
                entry = {"tables": {}}
                for ext in ("suf", "lcp", "llv", "bwt"):
                    pth = idx + "." + ext
                    entry["tables"][ext] = {"md5": md5(pth), "bytes": os.path.getsize(pth)}
                with open(idx + ".prj") as f:
                    entry["prj"] = f.read()
            variants["%s|%s|%d" % (name, d, int(mir))] = entry
    with open(os.path.join(OUT, "golden_variants.json"), "w") as f:
        json.dump(variants, f, indent=1, sort_keys=True)
    with open(os.path.join(OUT, "golden.json"), "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
