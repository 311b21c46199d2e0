"""Host-side C layer (libgtamd_host.so): FASTA encoder, sequence statistics
and the .prj writer against the reference's golden data and the oracle.  No
GPU needed; the tool function itself is exercised in test_cli_gpu.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle_util as ou
from genometools_amd import _lib
from genometools_amd._lib import EsaStats

GOLDEN = ou.golden()
with open(os.path.join(ou.GOLDEN_DIR, "golden_multi.json")) as _f:
    import json as _json
    MULTI = _json.load(_f)
HOST_DIR = os.path.join(_lib.HERE, "csrc", "host")
HOST_LIB = os.path.join(_lib.HERE, "libgtamd_host.so")


class SeqStats(ctypes.Structure):
    _fields_ = ou.SeqStats._fields_


class EncInfo(ctypes.Structure):           # gtamd_encinfo, include/gtamd_host.h
    _fields_ = [("originaldistribution", ctypes.c_uint64 * 256),
                ("filelengthtab", ctypes.c_void_p), ("numfiles", ctypes.c_size_t),
                ("exceptioncharacters", ctypes.c_uint64),
                ("realexceptionranges", ctypes.c_uint64)]


@pytest.fixture(scope="module")
def host():
    _lib.build_library()
    subprocess.run(["make", "-C", HOST_DIR], check=True, stdout=subprocess.DEVNULL)
    _lib.load()   # libgtamd_esa.so first (dependency)
    L = ctypes.CDLL(HOST_LIB)
    P = ctypes.c_void_p
    L.gtamd_encode_files.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_size_t,
                                     ctypes.c_int, ctypes.POINTER(P),
                                     ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p,
                                     ctypes.c_size_t]
    L.gtamd_encode_files_desc.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_size_t,
                                          ctypes.c_int, ctypes.POINTER(P),
                                          ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(P),
                                          ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p,
                                          ctypes.c_size_t]
    L.gtamd_encode_files_info.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_size_t,
                                          ctypes.c_int, ctypes.POINTER(P),
                                          ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(P),
                                          ctypes.POINTER(ctypes.c_uint64),
                                          ctypes.POINTER(EncInfo), ctypes.c_char_p,
                                          ctypes.c_size_t]
    L.gtamd_encinfo_free.argtypes = [ctypes.POINTER(EncInfo)]
    L.gtamd_write_esq.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p),
                                  ctypes.c_size_t, P, ctypes.c_uint64, ctypes.c_int,
                                  ctypes.POINTER(EncInfo), ctypes.c_int, ctypes.c_char_p,
                                  ctypes.c_size_t]
    L.gtamd_write_esq_sat.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p),
                                      ctypes.c_size_t, P, ctypes.c_uint64, ctypes.c_int,
                                      ctypes.POINTER(EncInfo), ctypes.c_int, ctypes.c_char_p,
                                      ctypes.POINTER(SeqStats), ctypes.c_char_p,
                                      ctypes.c_size_t]
    L.gtamd_read_esq.argtypes = [ctypes.c_char_p, ctypes.POINTER(P),
                                 ctypes.POINTER(ctypes.c_uint64),
                                 ctypes.POINTER(ctypes.c_int), ctypes.POINTER(SeqStats),
                                 ctypes.c_char_p, ctypes.c_size_t]
    L.gtamd_write_des_sds.argtypes = [ctypes.c_char_p, P, ctypes.c_uint64, ctypes.c_int,
                                      ctypes.c_int]
    L.gtamd_write_md5.argtypes = [ctypes.c_char_p, P, ctypes.c_uint64, ctypes.c_int]
    L.gtamd_sequence_stats.argtypes = [P, ctypes.c_uint64, ctypes.c_uint32,
                                       ctypes.POINTER(SeqStats)]
    L.gtamd_write_prj.argtypes = [ctypes.c_char_p, ctypes.POINTER(SeqStats),
                                  ctypes.POINTER(EsaStats), ctypes.c_int,
                                  ctypes.c_int, ctypes.c_int]
    L.gtamd_apply_readmode.argtypes = [P, ctypes.c_uint64, ctypes.c_int]
    L.gtamd_mirror.restype = P
    L.gtamd_mirror.argtypes = [P, ctypes.c_uint64]
    L.gtamd_seqstats_mirror.argtypes = [ctypes.POINTER(SeqStats), ctypes.c_int]
    L.gtamd_suffixerator.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                     ctypes.c_char_p, ctypes.c_size_t]
    return L


def _encode(host, paths, protein=False):
    arr = (ctypes.c_char_p * len(paths))(*[p.encode() for p in paths])
    ptr, n = ctypes.c_void_p(), ctypes.c_uint64()
    err = ctypes.create_string_buffer(2048)
    rc = host.gtamd_encode_files(arr, len(paths), int(protein), ctypes.byref(ptr),
                                 ctypes.byref(n), err, 2048)
    if rc != 0:
        raise ValueError(err.value.decode())
    enc = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)),
                                shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint8)
    ctypes.CDLL(None).free(ptr)
    return enc


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_encoder_and_prj_match_reference(host, name, tmp_path):
    e = GOLDEN[name]
    protein = e["alphabet"] == "protein"
    enc = _encode(host, [ou.fixture_path(name)], protein)
    assert np.array_equal(enc, ou.encode_fasta(ou.fixture_path(name), protein))
    prj = dict(l.split("=") for l in e["prj"].splitlines())
    assert enc.size == int(prj["totallength"])
    ss = SeqStats()
    host.gtamd_sequence_stats(enc.ctypes.data, enc.size, 20 if protein else 4,
                              ctypes.byref(ss))
    for key in ("specialcharacters", "specialranges", "realspecialranges",
                "lengthofspecialprefix", "lengthofspecialsuffix", "wildcards",
                "wildcardranges", "realwildcardranges", "lengthofwildcardprefix",
                "lengthofwildcardsuffix", "numofsequences"):
        assert getattr(ss, key) == int(prj[key]), key
    # the .prj writer, fed with the table statistics the oracle derives
    ora = ou.esa(enc, 20 if protein else 4)["stats"]
    es = EsaStats(totallength=enc.size, numberofallsortedsuffixes=enc.size + 1,
                  longest=ora["longest"], largelcpvalues=ora["largelcpvalues"],
                  maxbranchdepth=ora["maxbranchdepth"], lcptabsum=int(ora["lcptabsum"]),
                  prefixlength=ora["prefixlength"])
    out = str(tmp_path / "x.prj")
    assert host.gtamd_write_prj(out.encode(), ctypes.byref(ss), ctypes.byref(es), 1, 0, 0) == 0
    with open(out) as f:
        assert f.read() == e["prj"]


def test_multiple_files_are_joined_by_separators(host):
    # Small.fna + Verysmall.fna -> 151 + 1 + 8 = 160
    a = _encode(host, [ou.fixture_path("Small.fna")])
    b = _encode(host, [ou.fixture_path("Verysmall.fna")])
    ab = _encode(host, [ou.fixture_path("Small.fna"), ou.fixture_path("Verysmall.fna")])
    assert ab.size == a.size + 1 + b.size
    assert np.array_equal(ab, np.concatenate([a, [255], b]))


def test_encoder_errors_use_the_reference_wording(host, tmp_path):
    p = tmp_path / "bad.fna"
    p.write_text(">a\nACGT\nACXT\n")
    with pytest.raises(ValueError, match=r"illegal character 'X': file \".*bad.fna\", line 3"):
        _encode(host, [str(p)])
    p.write_text(">a\n>b\nACGT\n")
    with pytest.raises(ValueError, match="bad.fna' contains an empty sequence"):
        _encode(host, [str(p)])
    p.write_text(">a\nacgt\n>b\n")
    with pytest.raises(ValueError, match="contains an empty sequence"):
        _encode(host, [str(p)])
    p.write_text(">p\nlvif\n")     # lower-case protein letters are illegal
    with pytest.raises(ValueError, match="illegal character 'l'"):
        _encode(host, [str(p)], protein=True)
    with pytest.raises(ValueError, match="cannot open file"):
        _encode(host, [str(tmp_path / "missing.fna")])


def test_tool_argument_errors(host):
    def run(*args):
        argv = (ctypes.c_char_p * (len(args) + 1))(b"suffixerator", *[a.encode() for a in args])
        err = ctypes.create_string_buffer(2048)
        rc = host.gtamd_suffixerator(len(args) + 1, argv, err, 2048)
        return rc, err.value.decode()
    assert run("-suf") == (-1, 'either option "-db" or option "-ii" is mandatory')
    rc, msg = run("-db", "a.fna", "b.fna", "-suf")
    assert rc == -1 and "option -indexname is mandatory" in msg
    rc, msg = run("-db", ou.fixture_path("Atinsert.fna"), "-dir", "sideways", "-suf")
    assert rc == -1 and "must be fwd or rev or cpl or rcl" in msg
    rc, msg = run("-protein", "-db", ou.fixture_path("sw100K1.fsa"), "-dir", "rcl", "-suf")
    assert (rc, msg) == (-1, "option -rcl only can be used for DNA alphabets")
    rc, msg = run("-db", ou.fixture_path("Atinsert.fna"), "-frobnicate")
    assert rc == -1 and "unknown option" in msg


VARIANTS = __import__("json").load(open(os.path.join(ou.GOLDEN_DIR, "golden_variants.json")))


@pytest.mark.parametrize("key", sorted(VARIANTS))
def test_readmodes_and_mirror_match_reference(host, key, tmp_path):
    """-dir rev|cpl|rcl and -mirrored (testsuite/gt_suffixerator_include.rb:17-56,
    464-487): the host layer's sequence transforms + statistics, with the
    tables the oracle derives from the transformed sequence, reproduce the
    reference's files"""
    import hashlib
    name, d, mir = key.split("|")
    e = VARIANTS[key]
    readmode = {"fwd": 0, "rev": 1, "cpl": 2, "rcl": 3}[d]
    enc = _encode(host, [ou.fixture_path(name)])
    ss = SeqStats()
    host.gtamd_sequence_stats(enc.ctypes.data, enc.size, 4, ctypes.byref(ss))
    if mir == "1":
        ptr = host.gtamd_mirror(enc.ctypes.data, enc.size)
        host.gtamd_seqstats_mirror(ctypes.byref(ss), int(enc.size > 0 and enc[-1] == 254))
        m = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)),
                                  shape=(2 * enc.size + 1,)).copy()
        ctypes.CDLL(None).free(ctypes.c_void_p(ptr))
        enc = m
    enc = np.ascontiguousarray(enc)
    host.gtamd_apply_readmode(enc.ctypes.data, enc.size, readmode)
    ora = ou.esa(enc, 4)
    for ext in ("suf", "lcp", "llv", "bwt"):
        assert hashlib.md5(np.ascontiguousarray(ora[ext]).tobytes()).hexdigest() == e["tables"][ext]["md5"], ext
    st = ora["stats"]
    es = EsaStats(totallength=enc.size, numberofallsortedsuffixes=enc.size + 1,
                  longest=st["longest"], largelcpvalues=st["largelcpvalues"],
                  maxbranchdepth=st["maxbranchdepth"], lcptabsum=int(st["lcptabsum"]),
                  prefixlength=st["prefixlength"])
    out = str(tmp_path / "x.prj")
    assert host.gtamd_write_prj(out.encode(), ctypes.byref(ss), ctypes.byref(es), 1,
                                readmode, int(mir)) == 0
    with open(out) as f:
        assert f.read() == e["prj"]


def test_fastq_errors_use_the_reference_wording(host, tmp_path):
    """messages of src/core/seq_iterator_fastq.c, checked against the reference
    binary when this test was written"""
    p = tmp_path / "e.fastq"
    cases = [("@a\nACGT\n+\nIIII\nX\n", "'@' expected, 'X' encountered instead in line 5"),
             ("@a\nACGT\n+\nII", r"lengths of character sequence and qualities sequence differ \(2 <-> 4\)"),
             ("@a\n\n+\n\n", "empty sequence given in file '.*e.fastq', line 2"),
             ("@a\nACGT\n+b\nIIII\n", "sequence description 'a' is not equal to qualities description 'b' in line 3"),
             ("@a\nACGT\n+\nIIIII\n", "qualities string of sequence length 4 is not ended by newline in file '.*e.fastq', line 4"),
             ("@a\nACXT\n+\nIIII\n", "illegal character 'X': file \".*e.fastq\"")]
    for text, pattern in cases:
        p.write_text(text)
        with pytest.raises(ValueError, match=pattern):
            _encode(host, [str(p)])
    p.write_text("@r1\nACGT\nAC\n+r1\nII\nII@+\n@r2\nNNA\n+\n@@@\n")
    enc = _encode(host, [str(p)])
    assert enc.tolist() == [0, 1, 2, 3, 0, 1, 255, 254, 254, 0]


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_des_sds_md5_files_match_reference(host, name, tmp_path):
    """INDEX.des / .sds / .md5 as the reference's encoder writes them"""
    import hashlib
    e = GOLDEN[name]
    protein = e["alphabet"] == "protein"
    arr = (ctypes.c_char_p * 1)(ou.fixture_path(name).encode())
    ptr, n = ctypes.c_void_p(), ctypes.c_uint64()
    dptr, dlen = ctypes.c_void_p(), ctypes.c_uint64()
    err = ctypes.create_string_buffer(2048)
    assert host.gtamd_encode_files_desc(arr, 1, int(protein), ctypes.byref(ptr),
                                        ctypes.byref(n), ctypes.byref(dptr),
                                        ctypes.byref(dlen), err, 2048) == 0, err.value
    idx = str(tmp_path / "idx")
    assert host.gtamd_write_des_sds(idx.encode(), dptr, dlen.value, 1, 1) == 0
    assert host.gtamd_write_md5(idx.encode(), ptr, n.value, int(protein)) == 0
    libc = ctypes.CDLL(None)
    libc.free(ptr)
    libc.free(dptr)
    for ext in ("des", "sds", "md5"):
        with open(idx + "." + ext, "rb") as f:
            raw = f.read()
        assert len(raw) == e["seqfiles"][ext]["bytes"], ext
        assert hashlib.md5(raw).hexdigest() == e["seqfiles"][ext]["md5"], ext


def _write_all_seqfiles(host, paths, protein, idx, write_ssp=1):
    """encode paths and write every sequence-side file of index idx; INDEX.esq
    stores the input names as typed -- bare file names, as make_golden.py ran
    the reference"""
    arr = (ctypes.c_char_p * len(paths))(*[p.encode() for p in paths])
    names = (ctypes.c_char_p * len(paths))(*[os.path.basename(p).encode() for p in paths])
    ptr, n = ctypes.c_void_p(), ctypes.c_uint64()
    dptr, dlen = ctypes.c_void_p(), ctypes.c_uint64()
    info = EncInfo()
    err = ctypes.create_string_buffer(2048)
    assert host.gtamd_encode_files_info(arr, len(paths), int(protein), ctypes.byref(ptr),
                                        ctypes.byref(n), ctypes.byref(dptr),
                                        ctypes.byref(dlen), ctypes.byref(info),
                                        err, 2048) == 0, err.value
    assert host.gtamd_write_esq(idx.encode(), names, len(paths), ptr, n.value,
                                int(protein), ctypes.byref(info), write_ssp,
                                err, 2048) == 0, err.value
    assert host.gtamd_write_des_sds(idx.encode(), dptr, dlen.value, 1, 1) == 0
    assert host.gtamd_write_md5(idx.encode(), ptr, n.value, int(protein)) == 0
    host.gtamd_encinfo_free(ctypes.byref(info))
    libc = ctypes.CDLL(None)
    libc.free(ptr)
    libc.free(dptr)


def _check_seqfiles(idx, expected):
    import hashlib
    for ext in ("des", "sds", "md5", "esq", "ssp"):
        assert os.path.exists(idx + "." + ext) == (ext in expected), ext
        if ext in expected:
            with open(idx + "." + ext, "rb") as f:
                raw = f.read()
            assert len(raw) == expected[ext]["bytes"], ext
            assert hashlib.md5(raw).hexdigest() == expected[ext]["md5"], ext


# access type (word 2 of INDEX.esq) the reference chose for some fixtures: all
# seven layouts the writer knows are pinned by at least one of them
ACCESS_TYPES = {"sw100K1.fsa": 1, "extra/protein_long_x.faa": 1, "Reads1.fna": 2,
                "Atinsert.fna": 3, "Duplicate.fna": 4, "RandomN.fna": 5,
                "extra/uint32_tables.fna": 6, "extra/equal_length_one_n.fna": 5}


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_esq_and_ssp_files_match_reference(host, name, tmp_path):
    """INDEX.esq (header, access type, packed sequence, wildcard tables) and
    INDEX.ssp byte for byte as the reference's encoder writes them"""
    e = GOLDEN[name]
    idx = str(tmp_path / "idx")
    _write_all_seqfiles(host, [ou.fixture_path(name)], e["alphabet"] == "protein", idx)
    _check_seqfiles(idx, e["seqfiles"])
    if name in ACCESS_TYPES:
        with open(idx + ".esq", "rb") as f:
            words = np.frombuffer(f.read(32), dtype="<u8")
        assert words[2] == ACCESS_TYPES[name] and words[1] == 3


@pytest.mark.parametrize("key", sorted(MULTI))
def test_several_input_files_match_reference(host, key, tmp_path):
    """file length table, separators between files, the FASTQ reader's
    accounting across its 8192-symbol buffer"""
    e = MULTI[key]
    idx = str(tmp_path / "idx")
    paths = [os.path.join(ou.GOLDEN_DIR, "multi", f) for f in e["files"]]
    _write_all_seqfiles(host, paths, False, idx)
    _check_seqfiles(idx, e["seqfiles"])
    enc = _encode(host, paths)
    prj = dict(l.split("=") for l in e["prj"].splitlines())
    ss = SeqStats()
    host.gtamd_sequence_stats(enc.ctypes.data, enc.size, 4, ctypes.byref(ss))
    for k in ("totallength", "specialcharacters", "specialranges", "wildcardranges",
              "numofsequences"):
        assert getattr(ss, k) == int(prj[k]), k


def test_ssp_is_optional_unless_the_access_type_needs_it(host, tmp_path):
    # bit access (Atinsert): -ssp no drops the file; table access (Duplicate)
    # keeps it, the reference cannot find sequence boundaries without it
    # (src/core/encseq.c:4609-4619)
    idx = str(tmp_path / "a")
    _write_all_seqfiles(host, [ou.fixture_path("Atinsert.fna")], False, idx, write_ssp=0)
    assert not os.path.exists(idx + ".ssp")
    idx = str(tmp_path / "d")
    _write_all_seqfiles(host, [ou.fixture_path("Duplicate.fna")], False, idx, write_ssp=0)
    assert os.path.exists(idx + ".ssp")


def _read_esq(host, idx):
    ptr, n, protein = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_int()
    ss = SeqStats()
    err = ctypes.create_string_buffer(2048)
    if host.gtamd_read_esq(idx.encode(), ctypes.byref(ptr), ctypes.byref(n),
                           ctypes.byref(protein), ctypes.byref(ss), err, 2048) != 0:
        raise ValueError(err.value.decode())
    enc = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)),
                                shape=(n.value,)).copy()
    ctypes.CDLL(None).free(ptr)
    return enc, bool(protein.value), ss


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_esq_reader_returns_the_encoded_sequence(host, name, tmp_path):
    """-ii: INDEX.esq (+ .ssp) -- byte-identical to the reference's files, see
    test_esq_and_ssp_files_match_reference -- decodes to the symbols and the
    statistics the encoder produced, for every access type"""
    e = GOLDEN[name]
    protein = e["alphabet"] == "protein"
    idx = str(tmp_path / "idx")
    _write_all_seqfiles(host, [ou.fixture_path(name)], protein, idx)
    enc, was_protein, ss = _read_esq(host, idx)
    assert was_protein == protein
    assert np.array_equal(enc, _encode(host, [ou.fixture_path(name)], protein))
    prj = dict(l.split("=") for l in e["prj"].splitlines())
    for key in ("totallength", "specialcharacters", "specialranges", "realspecialranges",
                "lengthofspecialprefix", "lengthofspecialsuffix", "wildcards",
                "wildcardranges", "realwildcardranges", "lengthofwildcardprefix",
                "lengthofwildcardsuffix", "numofsequences"):
        assert getattr(ss, key) == int(prj[key]), key


def test_esq_reader_rejects_damaged_files(host, tmp_path):
    idx = str(tmp_path / "idx")
    with pytest.raises(ValueError, match="cannot open file '.*idx.esq'"):
        _read_esq(host, idx)
    _write_all_seqfiles(host, [ou.fixture_path("Duplicate.fna")], False, idx)
    good = open(idx + ".esq", "rb").read()
    open(idx + ".esq", "wb").write(good[:len(good) - 16])
    with pytest.raises(ValueError, match="truncated or inconsistent"):
        _read_esq(host, idx)
    open(idx + ".esq", "wb").write(good)
    os.unlink(idx + ".ssp")        # table access needs the separator positions
    with pytest.raises(ValueError, match="cannot open file '.*idx.ssp'"):
        _read_esq(host, idx)
    open(idx + ".esq", "wb").write(good[:8] + (2).to_bytes(8, "little") + good[16:])
    with pytest.raises(ValueError, match="unsupported format version"):
        _read_esq(host, idx)


def test_tool_ii_reuses_an_existing_index(host, tmp_path):
    """-ii INDEX without table options: only INDEX.prj is (re)written, from the
    statistics stored in INDEX.esq (src/match/sfx-opt.c:78-88,108-118)"""
    def run(*args):
        argv = (ctypes.c_char_p * (len(args) + 1))(b"suffixerator", *[a.encode() for a in args])
        err = ctypes.create_string_buffer(2048)
        return host.gtamd_suffixerator(len(args) + 1, argv, err, 2048), err.value.decode()
    idx = str(tmp_path / "first")
    assert run("-db", ou.fixture_path("Atinsert.fna"), "-indexname", idx) == (0, "")
    for ext in ("esq", "ssp", "des", "sds", "md5", "prj"):
        assert os.path.exists(idx + "." + ext), ext
    second = str(tmp_path / "second")
    assert run("-ii", idx, "-indexname", second) == (0, "")
    assert sorted(os.listdir(tmp_path)) == sorted(
        ["first." + e for e in ("esq", "ssp", "des", "sds", "md5", "prj")] + ["second.prj"])
    assert open(second + ".prj").read() == open(idx + ".prj").read()
    assert run("-ii", idx, "-db", "x.fna") == (-1, 'option "-db" and option "-ii" exclude each other')
    assert run("-ii", idx, "-dna") == (-1, 'option "-dna" and option "-ii" exclude each other')
    rc, msg = run("-ii", str(tmp_path / "nosuch"))
    assert rc == -1 and "cannot open file" in msg


ESQ_DIR = os.path.join(ou.GOLDEN_DIR, "esq")
REF_ESQ = sorted(f[:-4] for f in os.listdir(ESQ_DIR) if f.endswith(".esq"))


@pytest.mark.parametrize("stem", REF_ESQ)
def test_esq_reader_on_files_written_by_the_reference(host, stem, tmp_path):
    """tests/golden/esq/<input>.<sat>.esq[.ssp]: written by the reference with
    each access type forced (-sat direct|bytecompress|eqlen|bit|uchar|ushort|
    uint32, make_golden.py); the reader must return the encoder's symbols"""
    import shutil
    name, sat = stem.rsplit(".", 1)
    idx = str(tmp_path / "idx")
    for ext in ("esq", "ssp"):
        src = os.path.join(ESQ_DIR, "%s.%s" % (stem, ext))
        if os.path.exists(src):
            shutil.copyfile(src, idx + "." + ext)
    with open(idx + ".esq", "rb") as f:
        words = np.frombuffer(f.read(24), dtype="<u8")
    assert words[2] == {"direct": 0, "bytecompress": 1, "eqlen": 2, "bit": 3, "uchar": 4,
                        "ushort": 5, "uint32": 6}[sat]
    protein = name.endswith(".faa")
    fixture = ou.fixture_path(name if name in GOLDEN else "extra/" + name)
    enc, was_protein, ss = _read_esq(host, idx)
    assert was_protein == protein
    assert np.array_equal(enc, _encode(host, [fixture], protein))
    assert ss.totallength == enc.size
    assert ss.specialcharacters == int((enc >= 254).sum())
    assert ss.numofsequences == int((enc == 255).sum()) + 1


CLIPDESC = __import__("json").load(open(os.path.join(ou.GOLDEN_DIR, "golden_clipdesc.json")))


def _run_tool(host, *args, cwd=None):
    argv = (ctypes.c_char_p * (len(args) + 1))(b"suffixerator", *[a.encode() for a in args])
    err = ctypes.create_string_buffer(2048)
    old = os.getcwd()
    if cwd:
        os.chdir(cwd)
    try:
        return host.gtamd_suffixerator(len(args) + 1, argv, err, 2048), err.value.decode()
    finally:
        os.chdir(old)


@pytest.mark.parametrize("name", sorted(CLIPDESC))
def test_tool_clipdesc_matches_reference(host, name, tmp_path):
    """-clipdesc: INDEX.des/.sds with every description cut at its first
    white space (src/core/desc_buffer.c:63-80)"""
    import hashlib
    src = ou.fixture_path(name)
    idx = str(tmp_path / "idx")
    assert _run_tool(host, "-dna", "-clipdesc", "-indexname", idx, "-db",
                     os.path.basename(src), cwd=os.path.dirname(src)) == (0, "")
    for ext in ("des", "sds"):
        raw = open(idx + "." + ext, "rb").read()
        assert len(raw) == CLIPDESC[name][ext]["bytes"], ext
        assert hashlib.md5(raw).hexdigest() == CLIPDESC[name][ext]["md5"], ext


def test_tool_strategy_switches_are_accepted_and_output_switches_refused(host, tmp_path):
    """knobs of the reference's CPU algorithm do not change the files; options
    that would change them and are not built fail loudly"""
    src = ou.fixture_path("Atinsert.fna")
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    assert _run_tool(host, "-dna", "-indexname", a, "-db", src) == (0, "")
    assert _run_tool(host, "-dna", "-indexname", b, "-db", src, "-cmpcharbychar", "-dc", "32",
                     "-algbds", "3", "31", "80", "-maxwidthrealmedian", "1",
                     "-noshortreadsort", "-storespecialcodes", "yes", "-withradixsort",
                     "-iterscan", "no", "-parts", "2", "-memlimit", "1GB",
                     "-showprogress", "no", "-dccheck") == (0, "")
    assert open(a + ".prj").read() == open(b + ".prj").read()
    for opt in ("-plain", "-kys", "-lcpdist",
                "-compressedoutput", "-genomediff", "-sortmaxdepth", "-spmopt"):
        rc, msg = _run_tool(host, "-dna", "-indexname", a, "-db", src, opt)
        assert rc == -1 and msg == 'option "%s" is not supported by the MI355X engine' % opt


def _write_esq_forced(host, path, protein, idx, sat):
    arr = (ctypes.c_char_p * 1)(path.encode())
    names = (ctypes.c_char_p * 1)(os.path.basename(path).encode())
    ptr, n = ctypes.c_void_p(), ctypes.c_uint64()
    dptr, dlen = ctypes.c_void_p(), ctypes.c_uint64()
    info, ss = EncInfo(), SeqStats()
    err = ctypes.create_string_buffer(2048)
    assert host.gtamd_encode_files_info(arr, 1, int(protein), ctypes.byref(ptr), ctypes.byref(n),
                                        ctypes.byref(dptr), ctypes.byref(dlen),
                                        ctypes.byref(info), err, 2048) == 0, err.value
    rc = host.gtamd_write_esq_sat(idx.encode(), names, 1, ptr, n.value, int(protein),
                                  ctypes.byref(info), 1, sat.encode(), ctypes.byref(ss),
                                  err, 2048)
    host.gtamd_encinfo_free(ctypes.byref(info))
    libc = ctypes.CDLL(None)
    libc.free(ptr)
    libc.free(dptr)
    return rc, err.value.decode(), ss


@pytest.mark.parametrize("stem", REF_ESQ)
def test_forced_access_type_matches_reference(host, stem, tmp_path):
    """-sat TYPE: INDEX.esq / INDEX.ssp byte for byte as the reference writes
    them with that access type forced (tests/golden/esq/)"""
    name, sat = stem.rsplit(".", 1)
    fixture = ou.fixture_path(name if name in GOLDEN else "extra/" + name)
    idx = str(tmp_path / "idx")
    rc, msg, ss = _write_esq_forced(host, fixture, name.endswith(".faa"), idx, sat)
    assert rc == 0, msg
    for ext in ("esq", "ssp"):
        ref = os.path.join(ESQ_DIR, "%s.%s" % (stem, ext))
        assert os.path.exists(ref) == os.path.exists(idx + "." + ext), ext
        if os.path.exists(ref):
            assert open(idx + "." + ext, "rb").read() == open(ref, "rb").read(), ext
    # the statistics follow the forced table width (they are in the header too)
    hdr = np.frombuffer(open(idx + ".esq", "rb").read(168), dtype="<u8")
    assert (ss.specialranges, ss.wildcardranges) == (int(hdr[8]), int(hdr[13]))


def test_forced_access_type_errors_use_the_reference_wording(host, tmp_path):
    idx = str(tmp_path / "idx")
    dna, prot = ou.fixture_path("Atinsert.fna"), ou.fixture_path("extra/protein_specials.faa")
    rc, msg, _ = _write_esq_forced(host, dna, False, idx, "fast")
    assert rc == -1 and msg == ('Illegal argument "fast" to option -sat; must be one of the '
                                'following keywords: direct, bytecompress, eqlen, bit, uchar, '
                                'ushort, uint32')
    rc, msg, _ = _write_esq_forced(host, dna, False, idx, "bytecompress")
    assert rc == -1 and msg == ('illegal argument "bytecompress" to option -sat: cannot use '
                                'bytecompress on DNA sequences')
    rc, msg, _ = _write_esq_forced(host, dna, False, idx, "eqlen")
    assert rc == -1 and msg.startswith('illegal argument "eqlen" to option -sat: eqlen is only '
                                       'possible for DNA sequences, if all sequences are of equal')
    rc, msg, _ = _write_esq_forced(host, prot, True, idx, "bit")
    assert rc == -1 and msg == ('illegal argument "bit" to option -sat: as the sequence is not '
                                'DNA, you can choose bytecompress or direct')


SMAP = __import__("json").load(open(os.path.join(ou.GOLDEN_DIR, "golden_smap.json")))


@pytest.mark.parametrize("key", sorted(SMAP))
def test_tool_symbol_map_alphabets(host, key, tmp_path):
    """-smap FILE: alphabet from a symbol map; INDEX.esq carries the map text
    and alphabet type 2, the bit packing uses the alphabet's own width; all
    sequence-side files as the reference writes them"""
    import hashlib
    mapname, name = key.split("|")
    src = ou.fixture_path(name)
    idx = str(tmp_path / "idx")
    assert _run_tool(host, "-smap", os.path.join(ou.GOLDEN_DIR, "extra", mapname), "-indexname",
                     idx, "-db", os.path.basename(src), cwd=os.path.dirname(src)) == (0, "")
    e = SMAP[key]
    for ext in ("des", "sds", "md5", "esq", "ssp"):
        assert os.path.exists(idx + "." + ext) == (ext in e["seqfiles"]), ext
        if ext in e["seqfiles"]:
            raw = open(idx + "." + ext, "rb").read()
            assert hashlib.md5(raw).hexdigest() == e["seqfiles"][ext]["md5"], ext
    # and back: -ii reads the alphabet from the index
    assert _run_tool(host, "-ii", idx, "-indexname", str(tmp_path / "again")) == (0, "")
    want = dict(l.split("=") for l in e["prj"].splitlines())
    got = dict(l.split("=") for l in open(str(tmp_path / "again") + ".prj").read().splitlines())
    for k in ("totallength", "specialcharacters", "specialranges", "wildcards", "numofsequences",
              "prefixlength"):
        assert got[k] == want[k], k


def test_symbol_map_errors(host, tmp_path):
    src = ou.fixture_path("Atinsert.fna")
    idx = str(tmp_path / "idx")
    rc, msg = _run_tool(host, "-smap", str(tmp_path / "nothing"), "-indexname", idx, "-db", src)
    assert rc == -1 and "cannot open file" in msg
    bad = tmp_path / "bad.map"
    bad.write_text("aA\ncCa\nn\n")
    rc, msg = _run_tool(host, "-smap", str(bad), "-indexname", idx, "-db", src)
    assert rc == -1 and msg == "cannot map symbol 'a' to 1: it is already mapped to 0"
    assert _run_tool(host, "-smap", str(bad), "-dna", "-db", src) == \
        (-1, 'option "-smap" and option "-dna" exclude each other')
    # a protein-like map cannot be complemented
    rc, msg = _run_tool(host, "-smap", os.path.join(ou.GOLDEN_DIR, "extra", "prot5.map"), "-dir",
                        "rcl", "-indexname", idx, "-db", ou.fixture_path("extra/protein_specials.faa"))
    assert rc == -1 and msg == "option -rcl only can be used for DNA alphabets"


LOSSLESS = __import__("json").load(open(os.path.join(ou.GOLDEN_DIR, "golden_lossless.json")))


@pytest.mark.parametrize("name", sorted(LOSSLESS))
def test_tool_lossless_matches_reference(host, name, tmp_path):
    """-lossless: INDEX.ois (exception table), the exception counts in the
    header of INDEX.esq and INDEX.md5 over the original characters"""
    import hashlib
    src = ou.fixture_path(name)
    idx = str(tmp_path / "idx")
    flag = "-protein" if GOLDEN[name]["alphabet"] == "protein" else "-dna"
    assert _run_tool(host, flag, "-lossless", "-indexname", idx, "-db", os.path.basename(src),
                     cwd=os.path.dirname(src)) == (0, "")
    for ext, v in LOSSLESS[name].items():
        raw = open(idx + "." + ext, "rb").read()
        assert len(raw) == v["bytes"], ext
        assert hashlib.md5(raw).hexdigest() == v["md5"], ext
    assert os.path.exists(idx + ".ois")


def test_mergeesa_option_errors(host, tmp_path):
    """`gt dev mergeesa` (include/gtamd_host.h gtamd_mergeesa): the mandatory
    options and unreadable inputs are reported before any device is touched"""
    host.gtamd_mergeesa.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                    ctypes.c_char_p, ctypes.c_size_t]

    def run(*args):
        argv = (ctypes.c_char_p * (len(args) + 1))(b"mergeesa", *[a.encode() for a in args])
        err = ctypes.create_string_buffer(2048)
        return host.gtamd_mergeesa(len(args) + 1, argv, err, 2048), err.value.decode()

    assert run("-ii", "a", "b") == (-1, 'option "-indexname" is mandatory')
    assert run("-indexname", "x") == (-1, 'option "-ii" is mandatory')
    rc, msg = run("-indexname", str(tmp_path / "x"), "-ii", str(tmp_path / "nothere"))
    assert rc == -1 and "cannot open file" in msg and "nothere.suf" in msg
    rc, msg = run("-frobnicate")
    assert rc == -1 and "unknown option" in msg


def test_packedindex_trsuftab_option_errors(host, tmp_path):
    """`gt packedindex trsuftab` (include/gtamd_host.h gtamd_packedindex_trsuftab):
    options and unreadable projects are reported before any device is touched"""
    host.gtamd_packedindex_trsuftab.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                                ctypes.c_char_p, ctypes.c_size_t]

    def run(*args):
        argv = (ctypes.c_char_p * (len(args) + 1))(b"trsuftab", *[a.encode() for a in args])
        err = ctypes.create_string_buffer(2048)
        return host.gtamd_packedindex_trsuftab(len(args) + 1, argv, err, 2048), err.value.decode()

    assert run()[0] == -1
    rc, msg = run("-frobnicate", "x")
    assert rc == -1 and "unknown option" in msg
    rc, msg = run("-bsize")
    assert rc == -1 and "missing argument" in msg
    rc, msg = run("-bsize", "0", "x")
    assert rc == -1 and '"-bsize" must be an integer >= 1' in msg
    rc, msg = run("-ctxilog", "99", "x")
    assert rc == -1 and "between -2 and 63" in msg
    rc, msg = run("-sprank", str(tmp_path / "nothere"))
    assert rc == -1 and "nothere.prj" in msg
    rc, msg = run(str(tmp_path / "nothere"))
    assert rc == -1 and "nothere.prj" in msg
    rc, msg = run("a", "b")
    assert rc == -1 and "superfluous" in msg


def test_packedindex_mkindex_option_errors(host, tmp_path):
    """`gt packedindex mkindex`: the packed-index options are checked and the
    suffixerator's own checks apply before any device is touched"""
    host.gtamd_packedindex_mkindex.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                               ctypes.c_char_p, ctypes.c_size_t]

    def run(*args):
        argv = (ctypes.c_char_p * (len(args) + 1))(b"mkindex", *[a.encode() for a in args])
        err = ctypes.create_string_buffer(2048)
        return host.gtamd_packedindex_mkindex(len(args) + 1, argv, err, 2048), err.value.decode()

    assert run("-dna") == (-1, 'either option "-db" or option "-ii" is mandatory')
    rc, msg = run("-bwt", "-db", "x")
    assert rc == -1 and "unknown option: -bwt" in msg
    rc, msg = run("-blbuck", "0", "-db", "x")
    assert rc == -1 and '"-blbuck" must be an integer >= 1' in msg
    rc, msg = run("-locfreq", "-db", "x")
    assert rc == -1 and "non-negative integer" in msg
    rc, msg = run("-ctxilog", "-3", "-db", "x")
    assert rc == -1 and "between -2 and 63" in msg
    rc, msg = run("-dna", "-db", str(tmp_path / "nothere.fna"))
    assert rc == -1 and ("nothere.fna" in msg or "no HIP device" in msg)


def test_packedindex_mkctxmap_option_errors(host, tmp_path):
    host.gtamd_packedindex_mkctxmap.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                                ctypes.c_char_p, ctypes.c_size_t]

    def run(*args):
        argv = (ctypes.c_char_p * (len(args) + 1))(b"mkctxmap", *[a.encode() for a in args])
        err = ctypes.create_string_buffer(2048)
        return host.gtamd_packedindex_mkctxmap(len(args) + 1, argv, err, 2048), err.value.decode()

    assert run()[0] == -1
    rc, msg = run("-ctxilog")
    assert rc == -1 and "missing argument" in msg
    rc, msg = run("-ctxilog", "-2", "x")
    assert rc == -1 and ">= -1" in msg
    rc, msg = run(str(tmp_path / "nothere"))
    assert rc == -1 and "nothere.prj" in msg


def test_fastq_file_length_table_per_record_equals_per_symbol(host):
    """gtamd_fastq_filelengths books a record's symbols in one step; the reference's reader
    (src/core/sequence_buffer_fastq.c:42-191) goes symbol by symbol through its 8192-symbol
    output buffer -- restated here as it reads, with its quirks at the buffer boundaries --
    and the two must agree on records of every length around the buffer size, over
    several files"""
    class Rec(ctypes.Structure):
        _fields_ = [("seqlen", ctypes.c_uint64), ("desclen", ctypes.c_uint64), ("file", ctypes.c_size_t)]

    def per_symbol(recs, lastfile, nfiles):
        OUTBUF = 8192
        tab = [[0, 0] for _ in range(nfiles)]
        overflow, r, filenum, carry, complete = 0, 0, 0, 0, False
        while not complete:
            out = add = rd = 0
            if carry:
                out += 1; rd += 1; add += 1; carry = 0
            if overflow > 0:
                k = min(overflow, OUTBUF - out)
                out += k; add += k; rd += k; overflow -= k
                if overflow > 0:
                    continue
                out += 1; rd += 1
            while True:
                newfile = recs[r][2] if r < len(recs) else lastfile
                if filenum != newfile:
                    tab[filenum][0] += rd; tab[filenum][1] += add
                    rd = add = 0; filenum = newfile
                if r == len(recs):
                    complete = True; out -= 1; add -= 1
                    break
                for _ in range(recs[r][0]):
                    if out >= OUTBUF:
                        overflow += 1
                    else:
                        out += 1; add += 1; rd += 1
                if overflow == 0:
                    if out >= OUTBUF:
                        carry = 1
                    else:
                        out += 1; add += 1
                rd += recs[r][1] + 1
                r += 1
                if out >= OUTBUF:
                    break
            tab[filenum][0] += rd; tab[filenum][1] += add
        return tab

    rng = np.random.default_rng(19)
    host.gtamd_fastq_filelengths.restype = None
    for trial in range(60):
        nfiles = int(rng.integers(1, 4))
        nrec = int(rng.integers(1, 40))
        lens = rng.choice([1, 2, 100, 4095, 8190, 8191, 8192, 8193, 16383, 16384, 16385, 30000], size=nrec)
        nfiles = min(nfiles, nrec)
        files = np.sort(np.concatenate([np.arange(nfiles), rng.integers(0, nfiles, size=nrec - nfiles)]))
        recs = [(int(l), int(rng.integers(0, 30)), int(f)) for l, f in zip(lens, files)]
        arr = (Rec * nrec)(*[Rec(*t) for t in recs])
        tab = (ctypes.c_uint64 * (2 * nfiles))()
        host.gtamd_fastq_filelengths(arr, nrec, nfiles - 1, tab)
        want = per_symbol(recs, nfiles - 1, nfiles)
        assert [list(tab[2 * f:2 * f + 2]) for f in range(nfiles)] == \
            [[v % (1 << 64) for v in row] for row in want], (trial, recs)
