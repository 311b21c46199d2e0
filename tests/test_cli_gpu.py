"""`gt-suffixerator-amd` (C host layer + HIP engine) writes the same files as
the reference's `gt suffixerator` on the reference's own fixtures."""
import hashlib
import os
import subprocess

import pytest

import oracle_util as ou
from genometools_amd import _lib

pytestmark = pytest.mark.gpu
GOLDEN = ou.golden()
CLI = os.path.join(_lib.HERE, "gt-suffixerator-amd")


@pytest.fixture(scope="module")
def cli(gpu):
    subprocess.run(["make", "-C", os.path.join(_lib.HERE, "csrc", "host")], check=True,
                   stdout=subprocess.DEVNULL)
    return CLI


@pytest.mark.parametrize("name", ["Atinsert.fna", "Duplicate.fna", "RandomN.fna",
                                  "TTTN.fna", "Verysmall.fna", "Reads1.fna",
                                  "Copysorttest.fna", "sw100K1.fsa"])
def test_cli_writes_reference_files(cli, name, tmp_path):
    e = GOLDEN[name]
    idx = str(tmp_path / "idx")
    # INDEX.esq stores the -db argument as typed: bare name, run from its directory
    subprocess.run([cli, "-" + e["alphabet"], "-suf", "-lcp", "-bwt", "-tis", "-des",
                    "-ssp", "-db", os.path.basename(ou.fixture_path(name)), "-indexname", idx],
                   check=True, cwd=os.path.dirname(ou.fixture_path(name)))
    for ext in ("suf", "lcp", "llv", "bwt"):
        with open(idx + "." + ext, "rb") as f:
            raw = f.read()
        assert len(raw) == e["tables"][ext]["bytes"], ext
        assert hashlib.md5(raw).hexdigest() == e["tables"][ext]["md5"], ext
    with open(idx + ".prj") as f:
        assert f.read() == e["prj"]
    # sequence-side files written by default (-des -sds -md5 -ssp yes), and
    # the encoded sequence itself
    for ext in ("des", "sds", "md5", "esq", "ssp"):
        assert os.path.exists(idx + "." + ext) == (ext in e["seqfiles"]), ext
        if ext in e["seqfiles"]:
            with open(idx + "." + ext, "rb") as f:
                assert hashlib.md5(f.read()).hexdigest() == e["seqfiles"][ext]["md5"], ext


@pytest.mark.parametrize("chunk", ["64", "4096", "100000"])
@pytest.mark.parametrize("suftabuint", [False, True])
def test_cli_tables_written_in_many_pieces(cli, chunk, suftabuint, tmp_path, monkeypatch):
    """the table files leave through two staging buffers and a writer thread;
    GTAMD_TABLE_CHUNK shrinks the 64 MiB pieces so that a fixture's tables take
    hundreds of hand-overs (incl. the narrowing of the 32-bit suffix table)"""
    monkeypatch.setenv("GTAMD_TABLE_CHUNK", chunk)
    name = "Duplicate.fna"
    e = GOLDEN[name]
    idx = str(tmp_path / "idx")
    args = [cli, "-dna", "-suf", "-lcp", "-bwt", "-db", ou.fixture_path(name), "-indexname", idx]
    subprocess.run(args + (["-suftabuint"] if suftabuint else []), check=True)
    for ext in ("lcp", "llv", "bwt"):
        with open(idx + "." + ext, "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == e["tables"][ext]["md5"], ext
    with open(idx + ".suf", "rb") as f:
        raw = f.read()
    if suftabuint:
        import numpy as np
        wide = np.frombuffer(raw, dtype="<u4").astype("<u8").tobytes()
        assert hashlib.md5(wide).hexdigest() == e["tables"]["suf"]["md5"]
    else:
        assert hashlib.md5(raw).hexdigest() == e["tables"]["suf"]["md5"]


@pytest.mark.parametrize("stem", ["Duplicate.fna.ushort", "TTTN.fna.bit",
                                  "Atinsert_seqrange_3-7.fna.uint32", "Reads1.fna.eqlen",
                                  "protein_long_x.faa.bytecompress", "gt_in_line.fna.direct"])
def test_cli_ii_builds_tables_from_a_reference_written_index(cli, stem, tmp_path):
    """-ii INDEX: the encoded sequence comes from an INDEX.esq the reference
    wrote (tests/golden/esq/, one per access type); tables and .prj must be
    those of a run from the FASTA file"""
    import shutil
    name = stem.rsplit(".", 1)[0]
    e = GOLDEN[name if name in GOLDEN else "extra/" + name]
    src = str(tmp_path / "in")
    for ext in ("esq", "ssp"):
        f = os.path.join(ou.GOLDEN_DIR, "esq", "%s.%s" % (stem, ext))
        if os.path.exists(f):
            shutil.copyfile(f, src + "." + ext)
    idx = str(tmp_path / "out")
    subprocess.run([cli, "-ii", src, "-suf", "-lcp", "-bwt", "-indexname", idx], check=True)
    for ext in ("suf", "lcp", "llv", "bwt"):
        with open(idx + "." + ext, "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == e["tables"][ext]["md5"], ext
    assert not os.path.exists(idx + ".esq") and not os.path.exists(idx + ".des")
    got = dict(l.split("=") for l in open(idx + ".prj").read().splitlines())
    want = dict(l.split("=") for l in e["prj"].splitlines())
    # a forced access type changes how ranges are stored, nothing else
    for k in ("specialranges", "wildcardranges"):
        got.pop(k), want.pop(k)
    assert got == want


def test_cli_fastq_input_and_switches(cli, tmp_path):
    e = GOLDEN["test10_multiline.fastq"]
    idx = str(tmp_path / "fq")
    subprocess.run([cli, "-dna", "-suf", "-lcp", "-bwt", "-des", "no", "-md5", "no", "-db",
                    ou.fixture_path("test10_multiline.fastq"), "-indexname", idx], check=True)
    for ext in ("suf", "lcp", "llv", "bwt"):
        with open(idx + "." + ext, "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == e["tables"][ext]["md5"], ext
    assert not os.path.exists(idx + ".des") and not os.path.exists(idx + ".md5")
    with open(idx + ".sds", "rb") as f:
        assert hashlib.md5(f.read()).hexdigest() == e["seqfiles"]["sds"]["md5"]


def test_cli_suf_only_and_options_without_effect(cli, tmp_path):
    e = GOLDEN["Atinsert.fna"]
    idx = str(tmp_path / "idx")
    subprocess.run([cli, "-dna", "-suf", "-parts", "3", "-dc", "64", "-v", "-db",
                    ou.fixture_path("Atinsert.fna"), "-indexname", idx], check=True,
                   stdout=subprocess.DEVNULL)
    with open(idx + ".suf", "rb") as f:
        assert hashlib.md5(f.read()).hexdigest() == e["tables"]["suf"]["md5"]
    assert not os.path.exists(idx + ".lcp")
    prj = dict(l.split("=") for l in open(idx + ".prj").read().splitlines())
    assert prj["averagelcp"] == "0.00" and prj["maxbranchdepth"] == "0"
    assert prj["longest"] == "2529"


def test_cli_error_exit_code_and_message(cli, tmp_path):
    p = tmp_path / "bad.fna"
    p.write_text(">a\nACGTX\n")
    r = subprocess.run([cli, "-dna", "-suf", "-db", str(p), "-indexname",
                        str(tmp_path / "i")], capture_output=True, text=True)
    assert r.returncode == 1
    assert r.stderr.startswith("gt suffixerator: error: illegal character 'X': file \"")


VARIANTS = __import__("json").load(open(os.path.join(ou.GOLDEN_DIR, "golden_variants.json")))


@pytest.mark.parametrize("key", sorted(VARIANTS))
def test_cli_readmodes_and_mirror(cli, key, tmp_path):
    name, d, mir = key.split("|")
    e = VARIANTS[key]
    idx = str(tmp_path / "idx")
    subprocess.run([cli, "-dna", "-suf", "-lcp", "-bwt", "-dir", d, "-db",
                    ou.fixture_path(name), "-indexname", idx] +
                   (["-mirrored"] if mir == "1" else []), check=True)
    for ext in ("suf", "lcp", "llv", "bwt"):
        with open(idx + "." + ext, "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == e["tables"][ext]["md5"], ext
    with open(idx + ".prj") as f:
        assert f.read() == e["prj"]


MULTI = __import__("json").load(open(os.path.join(ou.GOLDEN_DIR, "golden_multi.json")))


@pytest.mark.parametrize("key", sorted(MULTI))
@pytest.mark.parametrize("encoder", ["device", "host"])
def test_cli_several_input_files(cli, key, encoder, tmp_path):
    """-db with several files, read on the device (FASTA) or on the host
    (FASTQ always): every file the reference writes, byte for byte"""
    e = MULTI[key]
    idx = str(tmp_path / "idx")
    subprocess.run([cli, "-dna", "-suf", "-lcp", "-bwt", "-encoder", encoder, "-indexname", idx,
                    "-db"] + e["files"], check=True, cwd=os.path.join(ou.GOLDEN_DIR, "multi"))
    for ext in ("suf", "lcp", "llv", "bwt"):
        with open(idx + "." + ext, "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == e["tables"][ext]["md5"], ext
    with open(idx + ".prj") as f:
        assert f.read() == e["prj"]
    for ext in ("des", "sds", "md5", "esq", "ssp"):
        assert os.path.exists(idx + "." + ext) == (ext in e["seqfiles"]), ext
        if ext in e["seqfiles"]:
            with open(idx + "." + ext, "rb") as f:
                assert hashlib.md5(f.read()).hexdigest() == e["seqfiles"][ext]["md5"], ext


# FASTQ fixtures not in the four-line form: the device reader declines them
FASTQ_FOR_THE_HOST = {"test10_multiline.fastq", "test5_tricky.fastq"}


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_cli_device_and_host_reader_write_the_same_index(cli, name, tmp_path):
    """every fixture through both readers; the device path (default) is the one
    compared with the reference's sequence-side files.  FASTQ in the four-line
    form is read on the device, the rest of FASTQ on the host (the tool says
    which reader it was)"""
    e = GOLDEN[name]
    src = ou.fixture_path(name)
    out = {}
    for encoder in ("device", "host"):
        idx = str(tmp_path / encoder)
        r = subprocess.run([cli, "-" + e["alphabet"], "-suf", "-v", "-encoder", encoder, "-indexname",
                            idx, "-db", os.path.basename(src)], check=True, cwd=os.path.dirname(src),
                           stdout=subprocess.PIPE, text=True)
        used = "host" if encoder == "host" or name in FASTQ_FOR_THE_HOST else "device"
        assert "(%s reader)" % used in r.stdout
        out[encoder] = {ext: hashlib.md5(open(idx + "." + ext, "rb").read()).hexdigest()
                        for ext in ("suf", "prj", "des", "sds", "md5", "esq", "ssp")
                        if os.path.exists(idx + "." + ext)}
    assert out["device"] == out["host"]
    for ext, v in e["seqfiles"].items():
        assert out["device"][ext] == v["md5"], ext
    assert out["device"]["suf"] == e["tables"]["suf"]["md5"]


BCK = __import__("json").load(open(os.path.join(ou.GOLDEN_DIR, "golden_bck.json")))


@pytest.mark.parametrize("key", sorted(BCK))
def test_cli_bucket_table_and_32bit_suffix_table(cli, key, tmp_path):
    """-bck [-pl K] and -suftabuint: INDEX.bck and the narrow INDEX.suf as the
    reference writes them"""
    name, pl = key.split("|")
    e = BCK[key]
    src = ou.fixture_path(name)
    idx = str(tmp_path / "idx")
    subprocess.run([cli, "-" + GOLDEN[name]["alphabet"], "-suf", "-lcp", "-bwt", "-bck",
                    "-suftabuint", "-indexname", idx, "-db", os.path.basename(src)] +
                   (["-pl", pl] if pl != "0" else []), check=True, cwd=os.path.dirname(src))
    for ext, key2 in (("bck", "bck"), ("suf", "suf32")):
        with open(idx + "." + ext, "rb") as f:
            raw = f.read()
        assert len(raw) == e[key2]["bytes"], ext
        assert hashlib.md5(raw).hexdigest() == e[key2]["md5"], ext
    with open(idx + ".prj") as f:
        assert f.read() == e["prj"]


ESQ_DIR = os.path.join(ou.GOLDEN_DIR, "esq")


@pytest.mark.parametrize("stem", sorted(f[:-4] for f in os.listdir(ESQ_DIR) if f.endswith(".esq")))
def test_cli_forced_access_type(cli, stem, tmp_path):
    """-sat TYPE through the device reader: INDEX.esq/.ssp as the reference
    writes them with that access type (tests/golden/esq/), same tables"""
    name, sat = stem.rsplit(".", 1)
    key = name if name in GOLDEN else "extra/" + name
    src = ou.fixture_path(key)
    idx = str(tmp_path / "idx")
    subprocess.run([cli, "-" + GOLDEN[key]["alphabet"], "-suf", "-sat", sat, "-indexname", idx,
                    "-db", os.path.basename(src)], check=True, cwd=os.path.dirname(src))
    for ext in ("esq", "ssp"):
        ref = os.path.join(ESQ_DIR, "%s.%s" % (stem, ext))
        assert os.path.exists(ref) == os.path.exists(idx + "." + ext), ext
        if os.path.exists(ref):
            assert open(idx + "." + ext, "rb").read() == open(ref, "rb").read(), ext
    with open(idx + ".suf", "rb") as f:
        assert hashlib.md5(f.read()).hexdigest() == GOLDEN[key]["tables"]["suf"]["md5"]


SMAP = __import__("json").load(open(os.path.join(ou.GOLDEN_DIR, "golden_smap.json")))


@pytest.mark.parametrize("key", sorted(SMAP))
@pytest.mark.parametrize("encoder", ["device", "host"])
def test_cli_symbol_map_alphabets(cli, key, encoder, tmp_path):
    """-smap FILE (5- and 4-letter alphabets from tests/golden/extra/*.map): all
    tables, the bucket table and the sequence-side files"""
    mapname, name = key.split("|")
    e = SMAP[key]
    src = ou.fixture_path(name)
    idx = str(tmp_path / "idx")
    subprocess.run([cli, "-smap", os.path.join(ou.GOLDEN_DIR, "extra", mapname), "-suf", "-lcp",
                    "-bwt", "-bck", "-encoder", encoder, "-indexname", idx, "-db",
                    os.path.basename(src)], check=True, cwd=os.path.dirname(src))
    for ext in ("suf", "lcp", "llv", "bwt", "bck"):
        with open(idx + "." + ext, "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == e["tables"][ext]["md5"], ext
    with open(idx + ".prj") as f:
        assert f.read() == e["prj"]
    for ext, v in e["seqfiles"].items():
        with open(idx + "." + ext, "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == v["md5"], ext


REF_CHECK = os.path.join(os.path.dirname(_lib.HERE), "oracle", "_ref", "gt_ref_check")


def _reference_accepts(idx):
    """oracle/_ref/gt_ref_check (built from the reference's sources by
    oracle/Makefile.ref): gt_mapsuffixarray maps every file of the index, then
    gt_suftab_lightweightcheck, gt_lcptab_lightweightcheck (Manzini's lcp9 from
    the mapped suffix table against .lcp/.llv) and a .bwt check run over it"""
    assert os.path.exists(REF_CHECK), "build oracle/_ref first (python -c 'import __graft_entry__ as g; g.build()')"
    return subprocess.run([REF_CHECK, idx], capture_output=True, text=True)


@pytest.mark.parametrize("name,extra", [
    ("Atinsert.fna", []), ("Duplicate.fna", []), ("RandomN.fna", []), ("Reads1.fna", []),
    ("sw100K1.fsa", []), ("extra/uint32_tables.fna", []), ("extra/long_runs.fna", []),
    ("Atinsert.fna", ["-dir", "rcl"]), ("Duplicate.fna", ["-mirrored"]),
    ("Atinsert.fna", ["-sat", "direct"]), ("ebola-genomes.fna.gz", []),
])
def test_reference_loads_and_verifies_our_index(cli, name, extra, tmp_path):
    """the other direction of the drop-in: an index written here is mapped and
    checked by the reference's own code"""
    e = GOLDEN[name]
    src = ou.fixture_path(name)
    idx = str(tmp_path / "idx")
    subprocess.run([cli, "-" + e["alphabet"], "-suf", "-lcp", "-bwt", "-indexname", idx, "-db",
                    os.path.basename(src)] + extra, check=True, cwd=os.path.dirname(src))
    r = _reference_accepts(idx)
    assert r.returncode == 0, r.stderr + r.stdout
    prj = dict(l.split("=") for l in open(idx + ".prj").read().splitlines())
    assert r.stdout.startswith("ok totallength=%s " % prj["totallength"])


def test_reference_checker_rejects_a_damaged_index(cli, tmp_path):
    """(the checker does check: one wrong LCP value, one swapped pair of suffixes)"""
    src = ou.fixture_path("Atinsert.fna")
    idx = str(tmp_path / "idx")
    subprocess.run([cli, "-dna", "-suf", "-lcp", "-bwt", "-indexname", idx, "-db",
                    os.path.basename(src)], check=True, cwd=os.path.dirname(src))
    assert _reference_accepts(idx).returncode == 0
    lcp = bytearray(open(idx + ".lcp", "rb").read())
    lcp[1000] ^= 1
    open(idx + ".lcp", "wb").write(bytes(lcp))
    assert _reference_accepts(idx).returncode != 0
    lcp[1000] ^= 1
    open(idx + ".lcp", "wb").write(bytes(lcp))
    suf = bytearray(open(idx + ".suf", "rb").read())
    suf[8 * 500:8 * 501], suf[8 * 501:8 * 502] = suf[8 * 501:8 * 502], suf[8 * 500:8 * 501]
    open(idx + ".suf", "wb").write(bytes(suf))
    assert _reference_accepts(idx).returncode != 0


def test_reference_verifies_a_large_index_from_the_device_reader(cli, tmp_path):
    """8 Mbp human-like genome (separators, wildcard runs, repeats with LCPs in
    the thousands): FASTA read on the device, all tables, checked by the
    reference's mappers and lightweight checkers"""
    from genometools_amd import synth
    enc = synth.generate(synth.MODEL_HUMANLIKE_DNA, 43, 8_000_000)
    fasta = str(tmp_path / "h8m.fna")
    synth.write_fasta(fasta, enc)
    idx = str(tmp_path / "idx")
    subprocess.run([cli, "-dna", "-suf", "-lcp", "-bwt", "-bck", "-indexname", idx, "-db",
                    "h8m.fna"], check=True, cwd=str(tmp_path))
    r = _reference_accepts(idx)
    assert r.returncode == 0, r.stderr + r.stdout


LOSSLESS = __import__("json").load(open(os.path.join(ou.GOLDEN_DIR, "golden_lossless.json")))


@pytest.mark.parametrize("name", ["Atinsert.fna", "extra/lowercase_across_records.fna",
                                  "sw100K1.fsa"])
def test_cli_lossless_with_tables(cli, name, tmp_path):
    """-lossless (host reader) together with a table build: the exception table
    and sequence files of the reference, the usual tables, and the reference's
    loader still accepts the index"""
    e = GOLDEN[name]
    src = ou.fixture_path(name)
    idx = str(tmp_path / "idx")
    subprocess.run([cli, "-" + e["alphabet"], "-lossless", "-suf", "-lcp", "-bwt", "-indexname",
                    idx, "-db", os.path.basename(src)], check=True, cwd=os.path.dirname(src))
    for ext, v in LOSSLESS[name].items():
        with open(idx + "." + ext, "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == v["md5"], ext
    for ext in ("suf", "lcp", "llv", "bwt"):
        with open(idx + "." + ext, "rb") as f:
            assert hashlib.md5(f.read()).hexdigest() == e["tables"][ext]["md5"], ext
    assert _reference_accepts(idx).returncode == 0


@pytest.mark.parametrize("files", [["Atinsert.fna", "Duplicate.fna"],
                                   ["Random.fna", "RandomN.fna", "TTTN.fna", "Atinsert.fna"],
                                   ["sw100K1.fsa", "sw100K2.fsa"]])
def test_mergeesa_equals_suffixerator_over_all_files(cli, files, tmp_path):
    """the reference's own merge test (testsuite/gt_mergeesa_include.rb:1-22): one
    index per file, `gt dev mergeesa`, and `cmp` of .suf/.lcp/.llv with the index
    over all files at once"""
    protein = files[0].endswith(".fsa")
    kind = "-protein" if protein else "-dna"
    paths = [ou.fixture_path(f) for f in files]
    subprocess.run([cli, kind, "-suf", "-lcp", "-indexname", str(tmp_path / "all"), "-db"] + paths,
                   check=True)
    idx = []
    for k, p in enumerate(paths):
        idx.append(str(tmp_path / ("midx%d" % k)))
        subprocess.run([cli, kind, "-suf", "-lcp", "-indexname", idx[-1], "-db", p], check=True)
    out = subprocess.run([cli, "mergeesa", "-indexname", str(tmp_path / "midx-all"), "-ii"] + idx,
                         check=True, capture_output=True, text=True).stdout
    assert out.startswith("# storeindex=")
    for ext in ("suf", "lcp", "llv"):
        with open(str(tmp_path / "all") + "." + ext, "rb") as a, \
                open(str(tmp_path / "midx-all") + "." + ext, "rb") as b:
            assert a.read() == b.read(), ext
    # an input without tables is refused, as the reference refuses to map it
    os.remove(idx[0] + ".lcp")
    r = subprocess.run([cli, "mergeesa", "-indexname", str(tmp_path / "x"), "-ii"] + idx,
                       capture_output=True, text=True)
    assert r.returncode == 1 and "gt dev mergeesa: error: cannot open file" in r.stderr


@pytest.mark.parametrize("name", ["Atinsert.fna", "Duplicate.fna", "sw100K2.fsa"])
def test_packedindex_trsuftab_writes_the_reference_file(cli, name, tmp_path):
    """`gt packedindex trsuftab` on a project written by the suffixerator tool:
    INDEX.bdx equals the file the reference wrote (tests/golden/golden_pck.json)"""
    golden = ou.golden_pck()
    protein = name.endswith(".fsa")
    idx = str(tmp_path / "pidx")
    keys = sorted(k for k in golden if k.split("|")[0] == name and "mode=" not in k)
    last_dir = None
    for key in sorted(keys, key=lambda k: ou.parse_pck_key(k)[1].get("direction", "fwd")):
        _, kw = ou.parse_pck_key(key)
        direction = kw.get("direction", "fwd")
        if direction != last_dir:
            subprocess.run([cli, "-protein" if protein else "-dna", "-suf", "-bwt", "-indexname", idx,
                            "-dir", direction, "-db", ou.fixture_path(name)], check=True)
            last_dir = direction
        args = ["-bsize", str(kw["bsize"]), "-blbuck", str(kw["blbuck"]), "-locfreq", str(kw["locfreq"])]
        if kw["locbitmap"] is not None:
            args += ["-locbitmap", "yes" if kw["locbitmap"] else "no"]
        if kw.get("sprank"):
            args += ["-sprank"]
        if os.path.exists(idx + ".bdx"):
            os.remove(idx + ".bdx")
        out = subprocess.run([cli, "packedindex", "trsuftab", "-v"] + args + [idx], check=True,
                             capture_output=True, text=True).stdout
        assert "buckets" in out
        with open(idx + ".bdx", "rb") as f:
            raw = f.read()
        assert len(raw) == golden[key]["size"], key
        assert hashlib.md5(raw).hexdigest() == golden[key]["md5"], key
    # without the .bwt table the tool says what it needs
    os.remove(idx + ".bwt")
    r = subprocess.run([cli, "packedindex", "trsuftab", idx], capture_output=True, text=True)
    assert r.returncode == 1 and "gt packedindex trsuftab: error:" in r.stderr and ".bwt" in r.stderr


@pytest.mark.parametrize("name", ["Atinsert.fna", "TTTN.fna", "sw100K2.fsa"])
def test_packedindex_mkindex_writes_the_reference_files(cli, name, tmp_path):
    """`gt packedindex mkindex`: INDEX.bdx (statistics flavour) and INDEX.prj
    equal what the reference's construction writes; the sequence-side files are
    the suffixerator's"""
    golden = ou.golden_pck()
    protein = name.endswith(".fsa")
    kind = "-protein" if protein else "-dna"
    ref = str(tmp_path / "ref")
    subprocess.run([cli, kind, "-indexname", ref, "-db", ou.fixture_path(name)], check=True)
    for key in sorted(k for k in golden if k.split("|")[0] == name and "mode=mkindex" in k):
        _, kw = ou.parse_pck_key(key)
        idx = str(tmp_path / "mk")
        args = ["-bsize", str(kw["bsize"]), "-blbuck", str(kw["blbuck"]), "-locfreq", str(kw["locfreq"])]
        if kw["locbitmap"] is not None:
            args += ["-locbitmap", "yes" if kw["locbitmap"] else "no"]
        if kw.get("sprank"):
            args += ["-sprank"]
        if kw.get("direction"):
            args += ["-dir", kw["direction"]]
        subprocess.run([cli, "packedindex", "mkindex", kind, "-indexname", idx, "-db",
                        ou.fixture_path(name)] + args, check=True)
        with open(idx + ".bdx", "rb") as f:
            raw = f.read()
        assert len(raw) == golden[key]["size"], key
        assert hashlib.md5(raw).hexdigest() == golden[key]["md5"], key
        with open(idx + ".prj") as f:
            assert f.read() == golden[key]["prj"], key
        assert not os.path.exists(idx + ".suf") and not os.path.exists(idx + ".bwt")
        for ext in ("ssp", "des", "sds", "md5"):
            with open(idx + "." + ext, "rb") as a, open(ref + "." + ext, "rb") as b:
                assert a.read() == b.read(), ext
    # protein: block sizes above 3 fall back to 3 (src/match/sfx-run.c:389-393)
    if protein:
        subprocess.run([cli, "packedindex", "mkindex", kind, "-indexname", str(tmp_path / "a"),
                        "-db", ou.fixture_path(name)], check=True)
        subprocess.run([cli, "packedindex", "mkindex", kind, "-bsize", "3", "-indexname",
                        str(tmp_path / "b"), "-db", ou.fixture_path(name)], check=True)
        with open(str(tmp_path / "a.bdx"), "rb") as a, open(str(tmp_path / "b.bdx"), "rb") as b:
            assert a.read() == b.read()
    r = subprocess.run([cli, "packedindex", "mkindex", kind, "-suf", "-db", ou.fixture_path(name)],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "gt packedindex mkindex: error: unknown option: -suf" in r.stderr


def test_packedindex_context_maps_through_the_tools(cli, tmp_path):
    """-ctxilog of trsuftab / mkindex and `gt packedindex mkctxmap`: INDEX.<I>cxm equals
    the reference's file (tests/golden/golden_ctxmap.json)"""
    ctx = ou.golden_ctxmap()
    name = "Atinsert.fna"
    idx = str(tmp_path / "cidx")
    subprocess.run([cli, "-dna", "-suf", "-bwt", "-indexname", idx, "-db", ou.fixture_path(name)],
                   check=True)

    def check(path, key):
        with open(path, "rb") as f:
            raw = f.read()
        assert len(raw) == ctx[key]["size"] and hashlib.md5(raw).hexdigest() == ctx[key]["md5"], key

    for ilog in (-1, 0, 3):
        key = "%s|ctxilog=%d" % (name, ilog)
        used = ctx[key]["used"]
        subprocess.run([cli, "packedindex", "mkctxmap", "-ctxilog", str(ilog), idx], check=True)
        check("%s.%dcxm" % (idx, used), key)
        os.remove("%s.%dcxm" % (idx, used))
        subprocess.run([cli, "packedindex", "trsuftab", "-ctxilog", str(ilog), idx], check=True)
        check("%s.%dcxm" % (idx, used), key)
        os.remove("%s.%dcxm" % (idx, used))
        mk = str(tmp_path / ("mk%d" % (ilog + 1)))
        subprocess.run([cli, "packedindex", "mkindex", "-dna", "-ctxilog", str(ilog), "-indexname", mk,
                        "-db", ou.fixture_path(name)], check=True)
        check("%s.%dcxm" % (mk, used), key)
    # no locate information, no map (the reference builds it beside the locate marks)
    subprocess.run([cli, "packedindex", "trsuftab", "-locfreq", "0", "-ctxilog", "3", idx], check=True)
    assert not os.path.exists(idx + ".3cxm")
    r = subprocess.run([cli, "packedindex", "mkctxmap", "-ctxilog", "40", idx], capture_output=True, text=True)
    assert r.returncode == 1 and "gt packedindex mkctxmap: error:" in r.stderr


@pytest.mark.parametrize("name,gpus", [("Atinsert.fna", 3), ("Duplicate.fna", 2), ("RandomN.fna", 5),
                                       ("Verysmall.fna", 4), ("sw100K1.fsa", 3)])
def test_cli_gpus_writes_reference_files(cli, name, gpus, tmp_path):
    """-gpus R: the tables built in R lexicographic ranges, one engine context and
    one host thread per range, over the transport the library ships
    (gtamd_comm_threads_create: peer copies; here all parts on the one GPU) -- C all
    the way, no Python in the build -- and streamed to the files in part order:
    the reference's files, .prj included"""
    e = GOLDEN[name]
    idx = str(tmp_path / "idx")
    subprocess.run([cli, "-" + e["alphabet"], "-suf", "-lcp", "-bwt", "-gpus", str(gpus), "-db",
                    os.path.basename(ou.fixture_path(name)), "-indexname", idx],
                   check=True, cwd=os.path.dirname(ou.fixture_path(name)))
    for ext in ("suf", "lcp", "llv", "bwt"):
        with open(idx + "." + ext, "rb") as f:
            raw = f.read()
        assert len(raw) == e["tables"][ext]["bytes"], ext
        assert hashlib.md5(raw).hexdigest() == e["tables"][ext]["md5"], ext
    with open(idx + ".prj") as f:
        assert f.read() == e["prj"]


def test_cli_gpus_40mbp_equals_reference_tables(cli, tmp_path):
    """a 40 Mbp human-like sequence through the C tool in three parts (range filter +
    MSD sort per part, pairs, rank exchange over the thread transport) against the
    tables the reference's own engine wrote for it (tests/golden/golden_large.json)"""
    import json
    from genometools_amd import synth
    with open(os.path.join(ou.GOLDEN_DIR, "golden_large.json")) as f:
        e = json.load(f)["humanlike_40m"]
    fasta = str(tmp_path / "h40.fna")
    synth.write_fasta(fasta, synth.generate(getattr(synth, e["model"]), e["seed"], e["n"]))
    idx = str(tmp_path / "idx")
    out = subprocess.run([cli, "-dna", "-suf", "-lcp", "-bwt", "-gpus", "3", "-v", "-db", fasta,
                          "-indexname", idx], check=True, capture_output=True, text=True).stdout
    assert "tables built in 3 parts" in out
    for ext in ("suf", "lcp", "llv", "bwt"):
        h = hashlib.md5()
        with open(idx + "." + ext, "rb") as f:
            for blk in iter(lambda: f.read(1 << 22), b""):
                h.update(blk)
        assert os.path.getsize(idx + "." + ext) == e["tables"][ext]["bytes"], ext
        assert h.hexdigest() == e["tables"][ext]["md5"], ext
    prj = dict(l.split("=") for l in e["prj"].splitlines())
    got = dict(l.split("=") for l in open(idx + ".prj").read().splitlines())
    for k in ("totallength", "numberofallsortedsuffixes", "longest", "prefixlength", "largelcpvalues",
              "averagelcp", "maxbranchdepth", "specialcharacters", "numofsequences"):
        assert got[k] == prj[k], k


def test_cli_gpus_refuses_what_needs_the_whole_table(cli, tmp_path):
    r = subprocess.run([cli, "-dna", "-suf", "-bck", "-gpus", "2", "-db", ou.fixture_path("Atinsert.fna"),
                        "-indexname", str(tmp_path / "x")], capture_output=True, text=True)
    assert r.returncode == 1 and "-gpus" in r.stderr and "-bck" in r.stderr
    r = subprocess.run([cli, "-dna", "-suf", "-gpus", "0", "-db", ou.fixture_path("Atinsert.fna"),
                        "-indexname", str(tmp_path / "x")], capture_output=True, text=True)
    assert r.returncode == 1 and "1 to 128" in r.stderr


def test_cli_fastq_reads_at_size_device_and_host_reader(cli, tmp_path):
    """4 M symbols of the human-like model as 40 000 four-line reads of 100 with names and
    pseudo-random qualities ('@' and '+' among them) in two files: the device FASTQ reader
    (records across the 4096-byte tiles of its kernels, the file length table booked over
    hundreds of buffer fills of the reference's reader) and the host reader write the same
    files, tables included"""
    import numpy as np
    from genometools_amd import synth
    n, L = 4_000_000, 100
    enc = synth.generate(synth.MODEL_HUMANLIKE_DNA, 11, n)
    lut = np.frombuffer(b"ACGT" + b"N" * 252, dtype=np.uint8)
    sym = lut[enc].reshape(-1, L)
    rng = np.random.default_rng(3)
    names = []
    for part, rows in (("a", sym[:25_000]), ("b", sym[25_000:])):
        p = tmp_path / ("reads_%s.fastq" % part)
        with open(p, "wb") as f:
            for k, row in enumerate(rows):
                name = b"read.%s.%d length=%d" % (part.encode(), k, L)
                qual = bytes(rng.integers(33, 127, size=L, dtype=np.uint8))
                f.write(b"@" + name + b"\n" + row.tobytes() + b"\n+" + (name if k % 5 == 0 else b"") +
                        b"\n" + qual + b"\n")
        names.append(p.name)
    out = {}
    for encoder in ("device", "host"):
        idx = str(tmp_path / encoder)
        r = subprocess.run([cli, "-dna", "-suf", "-lcp", "-bwt", "-v", "-encoder", encoder, "-indexname",
                            idx, "-db"] + names, check=True, cwd=str(tmp_path), stdout=subprocess.PIPE,
                           text=True)
        assert "(%s reader)" % encoder in r.stdout
        out[encoder] = {ext: hashlib.md5(open(idx + "." + ext, "rb").read()).hexdigest()
                        for ext in ("suf", "lcp", "llv", "bwt", "prj", "des", "sds", "md5", "esq", "ssp")
                        if os.path.exists(idx + "." + ext)}
    assert out["device"] == out["host"] and len(out["device"]) >= 9
