#!/usr/bin/env python3
"""Generate tests/golden/golden_pck.json from the REFERENCE itself (run in the
dev container).

For fixtures of the reference's packed-index tests
(testsuite/gt_packedindex_include.rb:67-127: the "simple sequences", the protein
sample with -bsize 1) and some more of its suffixerator fixtures this builds
the suffix-array project with oracle/_ref/gt_ref_sfx and then INDEX.bdx with
oracle/_ref/gt_ref_pck -- the reference's `gt packedindex trsuftab`
construction (src/match/eis-bwtseq-construct.c:64-92) compiled from
/root/reference by oracle/Makefile.ref -- for several option sets, and stores
md5 + size of every INDEX.bdx.  Only data is stored: the inputs are the
fixtures already under tests/golden/fixtures, the outputs are digests.
"""
import hashlib
import json
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SFX = os.path.join(ROOT, "oracle", "_ref", "gt_ref_sfx")
PCK = os.path.join(ROOT, "oracle", "_ref", "gt_ref_pck")
OUT = os.path.join(ROOT, "tests", "golden")

# (bsize, blbuck, locfreq, locbitmap: None = option not given)
OPTION_SETS = [
    (8, 8, 16, None),      # the tool's defaults: locate information as counts
    (8, 8, 16, True),      # -locbitmap yes
    (8, 8, 16, False),
    (8, 8, 0, None),       # -locfreq 0 (testsuite: "w/o locate")
    (4, 4, 8, None),
    (3, 5, 7, True),
    (5, 2, 1, False),
    (1, 3, 4, None),
    (2, 1, 3, True),
    (10, 3, 32, None),
    (8, 8, 1, True),
    (6, 7, 1000000, None),
]
DNA = ["RandomN.fna", "Random.fna", "Atinsert.fna", "TTT-small.fna",
       "trna_glutamine.fna", "Random-Small.fna", "Duplicate.fna", "TTTN.fna",
       "Verysmall.fna", "Small.fna", "Smalldup.fna", "Reads1.fna",
       "Random159.fna", "Random160.fna", "Copysorttest.fna",
       "Atinsert_seqrange_3-7.fna", "Arabidopsis-C99826.fna",
       "extra/starts_ends_special.fna", "extra/long_runs.fna"]
PROTEIN = [("sw100K2.fsa", [(1, 8, 16, None), (2, 8, 16, None), (2, 3, 5, True), (1, 1, 0, None)]),
           ("extra/protein_specials.faa", [(1, 8, 16, None), (2, 4, 3, False), (3, 2, 16, True)])]


def md5(path):
    with open(path, "rb") as f:
        return hashlib.md5(f.read()).hexdigest()


def key(name, opts, mkindex=False, sprank=False, direction=None):
    b, k, f, bm = opts
    return "%s|bsize=%d|blbuck=%d|locfreq=%d|locbitmap=%s" % (
        name, b, k, f, {None: "auto", True: "yes", False: "no"}[bm]) + \
        ("|mode=mkindex" if mkindex else "") + ("|sprank=yes" if sprank else "") + \
        ("|dir=" + direction if direction else "")


def run_case(tmp, name, protein, opts, mkindex=False, sprank=False, direction=None):
    src = os.path.join(OUT, "fixtures", name) if not name.startswith("extra/") \
        else os.path.join(OUT, name)
    idx = os.path.join(tmp, "idx")
    for f in os.listdir(tmp):
        os.unlink(os.path.join(tmp, f))
    # (a mkindex run takes the direction itself: the encoded sequence is stored forward)
    subprocess.run([SFX, "-protein" if protein else "-dna", "-suf", "-bwt", "-db", src,
                    "-indexname", idx] + (["-dir", direction] if direction and not mkindex else []),
                   check=True, stdout=subprocess.DEVNULL)
    b, k, f, bm = opts
    cmd = [PCK, "-bsize", str(b), "-blbuck", str(k), "-locfreq", str(f)]
    if bm is not None:
        cmd += ["-locbitmap", "yes" if bm else "no"]
    if sprank:
        cmd.append("-sprank")
    if mkindex and direction:
        cmd += ["-dir", direction]
    if mkindex:
        # the construction of `gt packedindex mkindex`: BWT from the suffixerator
        # interface, with sequence statistics (src/match/sfx-run.c:369-425)
        cmd.append("-mkindex")
    out = subprocess.run(cmd + [idx], check=True, capture_output=True, text=True).stdout
    toggles = int(out.split("featureToggles=")[1].split()[0])
    e = {"md5": md5(idx + ".bdx"), "size": os.path.getsize(idx + ".bdx"),
         "featureToggles": toggles}
    if mkindex:
        with open(idx + ".prj") as f:
            e["prj"] = f.read()
    return e


def main():
    golden = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name in DNA:
            sets = OPTION_SETS if name in ("Atinsert.fna", "Duplicate.fna", "RandomN.fna",
                                           "TTTN.fna", "Verysmall.fna",
                                           "extra/starts_ends_special.fna") \
                else OPTION_SETS[:4]
            for opts in sets:
                golden[key(name, opts)] = run_case(tmp, name, False, opts)
        for name, sets in PROTEIN:
            for opts in sets:
                golden[key(name, opts)] = run_case(tmp, name, True, opts)
        # mkindex flavour (statistics): the suite's "simple sequences" with the
        # defaults and without locate information, the protein sample with -bsize 1
        for name in DNA:
            for opts in (OPTION_SETS[:4] if name != "Atinsert.fna" else OPTION_SETS):
                golden[key(name, opts, True)] = run_case(tmp, name, False, opts, True)
        for name, sets in PROTEIN:
            for opts in sets:
                golden[key(name, opts, True)] = run_case(tmp, name, True, opts, True)
        # -sprank (testsuite/gt_packedindex_include.rb:87-118; the option set of the
        # index searches, testsuite/gt_idxsearch_include.rb:67-68: -bsize 10 -locfreq 32)
        sp_sets = [(8, 8, 16, None), (8, 8, 16, True), (10, 8, 32, None), (3, 5, 7, False),
                   (4, 4, 1, True), (8, 8, 0, None)]
        for name in ["RandomN.fna", "Random.fna", "Atinsert.fna", "TTT-small.fna",
                     "trna_glutamine.fna", "Random-Small.fna", "Duplicate.fna", "TTTN.fna",
                     "Verysmall.fna", "Atinsert_seqrange_3-7.fna",
                     "extra/starts_ends_special.fna", "extra/long_runs.fna"]:
            for opts in sp_sets:
                for mk in (False, True):
                    golden[key(name, opts, mk, True)] = run_case(tmp, name, False, opts, mk, True)
        for opts in [(1, 8, 16, None), (2, 3, 5, True)]:
            for mk in (False, True):
                golden[key("sw100K2.fsa", opts, mk, True)] = run_case(tmp, "sw100K2.fsa", True, opts, mk, True)
        # projects read in another direction (testsuite/gt_packedindex_include.rb:97-108
        # "revcom sequence with sprank": -dir rev; the index searches build theirs with
        # -dir rev too): trsuftab on a project the suffixerator wrote with -dir
        for name in ["Atinsert.fna", "Duplicate.fna", "TTTN.fna"]:
            for direction in ("rev", "cpl", "rcl"):
                for opts, sp in (((8, 8, 16, None), False), ((8, 8, 16, None), True),
                                 ((10, 8, 32, None), True), ((3, 5, 7, True), False)):
                    golden[key(name, opts, False, sp, direction)] = \
                        run_case(tmp, name, False, opts, False, sp, direction)
        # ... and mkindex with -dir, among them the command line of the index searches
        # (testsuite/gt_idxsearch_include.rb:67-68: -sprank -bsize 10 -locfreq 32 -dir rev)
        # (rev only: with cpl / rcl the reference takes its statistics from the stored
        # sequence, not the complemented one it indexes, and stops on its own assertion,
        # src/match/eis-bwtseq.c:97)
        for name in ["Atinsert.fna", "Duplicate.fna"]:
            for direction in ("rev",):
                for opts, sp in (((10, 8, 32, None), True), ((8, 8, 16, None), False)):
                    golden[key(name, opts, True, sp, direction)] = \
                        run_case(tmp, name, False, opts, True, sp, direction)
        # context maps (-ctxilog I; testsuite/gt_packedindex_include.rb:50-57, 110-118):
        # md5 + size of INDEX.<I>cxm; keys "name|ctxilog=I[|dir=..]", "used" = the I
        # of the file name (-1: the automatic interval)
        ctx = {}
        for name, protein in [("Atinsert.fna", False), ("Duplicate.fna", False),
                              ("TTTN.fna", False), ("Verysmall.fna", False),
                              ("TTT-small.fna", False), ("Random.fna", False),
                              ("sw100K2.fsa", True)]:
            for direction in (None, "rev") if name == "Atinsert.fna" else (None,):
                for ilog in (-1, 0, 1, 3, 5):
                    src = os.path.join(OUT, "fixtures", name)
                    idx = os.path.join(tmp, "idx")
                    for f in os.listdir(tmp):
                        os.unlink(os.path.join(tmp, f))
                    subprocess.run([SFX, "-protein" if protein else "-dna", "-suf", "-bwt", "-db",
                                    src, "-indexname", idx] +
                                   (["-dir", direction] if direction else []),
                                   check=True, stdout=subprocess.DEVNULL)
                    r = subprocess.run([PCK, "-ctxilog", str(ilog)] + (["-bsize", "2"] if protein else []) + [idx],
                                       capture_output=True, text=True)
                    maps = [f for f in os.listdir(tmp) if f.endswith("cxm")]
                    if r.returncode != 0 or len(maps) != 1:
                        continue      # an interval the reference refuses for this length
                    used = int(maps[0].split(".")[-1][:-3])
                    k = "%s|ctxilog=%d" % (name, ilog) + ("|dir=" + direction if direction else "")
                    ctx[k] = {"md5": md5(os.path.join(tmp, maps[0])),
                              "size": os.path.getsize(os.path.join(tmp, maps[0])), "used": used}
        with open(os.path.join(OUT, "golden_ctxmap.json"), "w") as f:
            json.dump(ctx, f, indent=1, sort_keys=True)
        print("%d context-map goldens" % len(ctx))
    with open(os.path.join(OUT, "golden_pck.json"), "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)
    print("%d packed-index goldens" % len(golden))


if __name__ == "__main__":
    main()
