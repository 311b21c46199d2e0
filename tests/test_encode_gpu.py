"""Device FASTA encoder (csrc/esa_encode.hip, include/gtamd_encode.h) against
the C host reader -- itself pinned to the reference's encoder by the golden
.prj/.esq/.des files -- and against numpy restatements of the run statistics."""
import ctypes
import os

import numpy as np
import pytest

import oracle_util as ou
from genometools_amd import encode, synth
from genometools_amd._lib import EsaError
from test_host import EncInfo, host  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
GOLDEN = ou.golden()
FASTA = sorted(n for n in GOLDEN if not n.endswith(".fastq"))
MULTI_FASTA = [["part1.fna", "part2.fna"], ["part2.fna", "part1.fna", "part2.fna"]]


def _host_encode(host, paths, protein):
    """symbols, descriptions, original-character histogram and file lengths of
    the host reader"""
    arr = (ctypes.c_char_p * len(paths))(*[p.encode() for p in paths])
    ptr, n = ctypes.c_void_p(), ctypes.c_uint64()
    dptr, dlen = ctypes.c_void_p(), ctypes.c_uint64()
    info = EncInfo()
    err = ctypes.create_string_buffer(2048)
    if host.gtamd_encode_files_info(arr, len(paths), int(protein), ctypes.byref(ptr),
                                    ctypes.byref(n), ctypes.byref(dptr), ctypes.byref(dlen),
                                    ctypes.byref(info), err, 2048) != 0:
        raise ValueError(err.value.decode())
    enc = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)),
                                shape=(n.value,)).copy()
    desc = ctypes.string_at(dptr, dlen.value).split(b"\0")[:-1]
    fl = np.ctypeslib.as_array(ctypes.cast(info.filelengthtab, ctypes.POINTER(ctypes.c_uint64)),
                               shape=(2 * len(paths),)).copy().reshape(-1, 2)
    orig = list(info.originaldistribution)
    host.gtamd_encinfo_free(ctypes.byref(info))
    libc = ctypes.CDLL(None)
    libc.free(ptr)
    libc.free(dptr)
    return enc, desc, orig, [tuple(int(x) for x in row) for row in fl]


def _runs(mask):
    """lengths of the maximal runs of True"""
    m = np.concatenate([[False], mask, [False]]).astype(np.int8)
    d = np.diff(m)
    return np.flatnonzero(d == -1) - np.flatnonzero(d == 1)


def _expected_summary(enc, sigma):
    sp, wc = _runs(enc >= 254), _runs(enc == 254)
    seqs, nonsp = _runs(enc != 255), _runs(enc < 254)

    def lead(mask):
        return int(np.argmin(mask)) if not mask.all() else mask.size

    def pieces(r, maxv):
        return int(((r + maxv) // (maxv + 1)).sum())
    return {
        "totallength": enc.size, "numofsequences": int((enc == 255).sum()) + 1,
        "specialcharacters": int((enc >= 254).sum()), "realspecialranges": sp.size,
        "specialrangestab": [pieces(sp, 255), pieces(sp, 65535), sp.size],
        "lengthofspecialprefix": lead(enc >= 254),
        "lengthofspecialsuffix": lead((enc >= 254)[::-1]),
        "wildcards": int((enc == 254).sum()), "realwildcardranges": wc.size,
        "wildcardrangestab": [pieces(wc, 255), pieces(wc, 65535), wc.size],
        "lengthofwildcardprefix": lead(enc == 254),
        "lengthofwildcardsuffix": lead((enc == 254)[::-1]),
        "lengthoflongestnonspecial": int(nonsp.max()) if nonsp.size else 0,
        "minseqlen": int(seqs.min()), "maxseqlen": int(seqs.max()),
        "equallength": int(seqs.min() == seqs.max() and not (enc == 254).any()),
        "characterdistribution": [int((enc == c).sum()) for c in range(sigma)] + [0] * (32 - sigma),
    }


def _check(host, paths, protein):
    want_enc, want_desc, want_orig, want_fl = _host_encode(host, paths, protein)
    with encode.DeviceEncoder(protein=protein) as de:
        de.encode(paths)
        got = de.symbols()
        assert de.length == want_enc.size
        assert np.array_equal(got, want_enc)
        assert de.descriptions() == want_desc
        assert de.file_lengths() == want_fl
        s = de.summary()
    assert s["originaldistribution"] == want_orig
    exp = _expected_summary(want_enc, 20 if protein else 4)
    for k, v in exp.items():
        assert s[k] == v, k
    return s


@pytest.mark.parametrize("name", FASTA)
def test_device_encoder_matches_host_reader(gpu, host, name):
    e = GOLDEN[name]
    s = _check(host, [ou.fixture_path(name)], e["alphabet"] == "protein")
    # and the reference's own numbers (.prj written by gt suffixerator)
    prj = dict(l.split("=") for l in e["prj"].splitlines())
    for k in ("totallength", "specialcharacters", "realspecialranges", "wildcards",
              "realwildcardranges", "lengthofspecialprefix", "lengthofspecialsuffix",
              "lengthofwildcardprefix", "lengthofwildcardsuffix", "numofsequences"):
        assert s[k] == int(prj[k]), k
    assert int(prj["specialranges"]) in s["specialrangestab"]
    assert int(prj["wildcardranges"]) in s["wildcardrangestab"]


@pytest.mark.parametrize("files", MULTI_FASTA)
def test_device_encoder_joins_files(gpu, host, files):
    _check(host, [os.path.join(ou.GOLDEN_DIR, "multi", f) for f in files], False)


def test_device_encoder_state_machine_corner_cases(gpu, host, tmp_path):
    cases = {
        # no '>' before the first symbols; a description cut off by the end of file
        "headless.fna": b"ACGT\nAC\n>second\nGG\n>third no newline",
        # '>' in the middle of a line opens a description there
        "midline.fna": b">a\nACGT>b rest of line\nTTTT\n",
        # CRLF, blank lines, tabs and form feeds inside the sequence
        "blanks.fna": b">x y\r\nAC GT\r\n\r\n\tNN\x0cA\r\n>z\r\nT\r\n",
        # '>' inside a description does not open another one
        "gt_in_desc.fna": b">a > b >> c\nACGT\n>d\nA\n",
    }
    for name, raw in cases.items():
        p = tmp_path / name
        p.write_bytes(raw)
        if name == "headless.fna":
            with pytest.raises(EsaError, match="headless.fna' contains an empty sequence"):
                encode.DeviceEncoder().encode([str(p)])
            with pytest.raises(ValueError, match="contains an empty sequence"):
                _host_encode(host, [str(p)], False)
            p.write_bytes(raw + b"\nA")
        _check(host, [str(p)], False)
    # tile boundaries: descriptions and line ends around multiples of 4096
    rng = np.random.default_rng(7)
    for width in (4093, 4094, 4095, 4096, 4097, 8191):
        seq = "".join(rng.choice(list("ACGTN"), size=3 * width))
        p = tmp_path / ("w%d.fna" % width)
        p.write_text(">%s\n%s\n>%s\n%s" % ("d" * (width - 2), seq[:width], "e" * 5000,
                                           seq[width:]))
        _check(host, [str(p)], False)


def test_device_encoder_errors_use_the_reference_wording(gpu, tmp_path):
    p = tmp_path / "bad.fna"
    p.write_text(">a\nACGT\nACXT\n")
    with pytest.raises(EsaError, match=r"illegal character 'X': file \".*bad.fna\", line 3"):
        encode.DeviceEncoder().encode([str(p)])
    p.write_text(">a\n>b\nACGT\n")
    with pytest.raises(EsaError, match="bad.fna' contains an empty sequence"):
        encode.DeviceEncoder().encode([str(p)])
    p.write_text(">a\nacgt\n>b\n")
    with pytest.raises(EsaError, match="contains an empty sequence"):
        encode.DeviceEncoder().encode([str(p)])
    p.write_text("ACGT\nACGT\n")
    with pytest.raises(EsaError, match=r"no sequences in multiple fasta file\(s\) .*bad.fna"):
        encode.DeviceEncoder().encode([str(p)])
    p.write_text(">p\nlvif\n")
    with pytest.raises(EsaError, match="illegal character 'l'"):
        encode.DeviceEncoder(protein=True).encode([str(p)])
    # the first illegal byte of a large input, far from the start
    big = np.frombuffer(b">x\n" + b"ACGT" * 3_000_000 + b"\n", dtype=np.uint8).copy()
    big[9_000_001] = ord("!")
    big[11_000_000] = ord("?")
    q = tmp_path / "big.fna"
    big.tofile(q)
    with pytest.raises(EsaError, match=r"illegal character '!': file \".*big.fna\", line 2"):
        encode.DeviceEncoder().encode([str(q)])


def test_device_encoder_large_input_and_engine_handover(gpu, host, tmp_path):
    """64 Mbp human-like genome as a 70-column FASTA: same symbols and numbers
    as the host reader; the encoded sequence goes to the ESA engine without
    leaving the device"""
    from genometools_amd import esa
    n = 64_000_000
    enc = synth.generate(synth.MODEL_HUMANLIKE_DNA, 43, n)
    path = str(tmp_path / "h64m.fna")
    synth.write_fasta(path, enc)
    s = _check(host, [path], False)
    assert s["totallength"] == n
    with encode.DeviceEncoder() as de:
        de.encode([path])
        t = de.timing()
        print("device encoder: %.1f MB in %.2f ms (parse %.2f, statistics %.2f)"
              % (t["input_bytes"] / 1e6, t["total_ms"], t["parse_ms"], t["stats_ms"]))
        with esa.EsaEngine(n, 4) as eng:
            eng.set_sequence_device(de.device_pointer, de.length)
            eng.run(esa.WANT_SUF)
            suf = eng.table(esa.TAB_SUF)
    assert ou.check_suffix_array(enc, suf)[0] == 0


def test_capacity_is_checked(gpu, tmp_path):
    """every entry point that fills caller arrays takes their capacity and
    writes nothing when the section does not fit (round 1's fuzz campaign
    stopped on an overwritten heap block once: no entry point may be able to)"""
    enc = synth.generate(synth.MODEL_HUMANLIKE_DNA, 3, 100000)
    path = str(tmp_path / "cap.fna")
    synth.write_fasta(path, enc)
    lib = gpu
    with encode.DeviceEncoder() as de:
        de.encode([path])
        k = lib.gtamd_encoder_num_descriptions(de._enc)
        assert k == 24
        guard = 0xA5A5A5A5A5A5A5A5
        f = np.full(k, 7, dtype=np.uint32)
        a = np.full(k, guard, dtype=np.uint64)
        b = np.full(k, guard, dtype=np.uint64)
        assert lib.gtamd_encoder_get_descriptions(de._enc, f.ctypes.data, a.ctypes.data,
                                                  b.ctypes.data, k - 1) == -1
        assert b"do not fit" in lib.gtamd_esa_last_error()
        assert (a == guard).all() and (b == guard).all() and (f == 7).all()
        assert lib.gtamd_encoder_get_descriptions(de._enc, f.ctypes.data, a.ctypes.data,
                                                  b.ctypes.data, k) == 0
        s = de.summary()
        n = de.length
        words = np.full(2 + (n - 1) // 32, guard, dtype=np.uint64)
        assert lib.gtamd_encoder_pack_twobit(de._enc, 1, 0, words.ctypes.data, words.size - 1) == -1
        assert (words == guard).all()
        assert lib.gtamd_encoder_pack_twobit(de._enc, 1, 0, words.ctypes.data, words.size) == 0
        sb = np.full(1 + (n + 63) // 64, guard, dtype=np.uint64)
        assert lib.gtamd_encoder_pack_specialbits(de._enc, sb.ctypes.data, sb.size - 1) == -1
        assert (sb == guard).all()
        assert lib.gtamd_encoder_pack_specialbits(de._enc, sb.ctypes.data, sb.size) == 0
        pk = np.full((3 * n + 7) // 8, 0xA5, dtype=np.uint8)
        assert lib.gtamd_encoder_pack_bytecompress(de._enc, pk.ctypes.data, pk.size - 1) == -1
        assert (pk == 0xA5).all()
        runs = s["realwildcardranges"]
        ws = np.full(runs, guard, dtype=np.uint64)
        wl = np.full(runs, guard, dtype=np.uint64)
        assert lib.gtamd_encoder_get_wildcard_runs(de._enc, ws.ctypes.data, wl.ctypes.data,
                                                   runs - 1) == -1
        assert (ws == guard).all() and (wl == guard).all()
        assert lib.gtamd_encoder_get_wildcard_runs(de._enc, ws.ctypes.data, wl.ctypes.data,
                                                   runs) == 0
        assert int(wl.sum()) == s["wildcards"]
        sep = np.full(23, guard, dtype=np.uint64)
        assert lib.gtamd_encoder_get_separators(de._enc, sep.ctypes.data, 22) == -1
        assert (sep == guard).all()
        assert lib.gtamd_encoder_get_separators(de._enc, sep.ctypes.data, 23) == 0
        assert np.array_equal(sep, np.flatnonzero(enc == 255).astype(np.uint64))


# ---- FASTQ on the device -----------------------------------------------------
def _device_encode_via_host(host, paths, protein):
    """gtamd_device_encode_files (include/gtamd_host.h): the host layer's use of
    the device reader, with the file length table booked as the reference's FASTQ
    reader does; returns None if the device reader declined"""
    arr = (ctypes.c_char_p * len(paths))(*[p.encode() for p in paths])
    de = ctypes.c_void_p()
    dptr, dlen = ctypes.c_void_p(), ctypes.c_uint64()
    info = EncInfo()
    err = ctypes.create_string_buffer(2048)
    host.gtamd_device_encode_files.restype = ctypes.c_int
    rc = host.gtamd_device_encode_files(arr, len(paths), int(protein), ctypes.byref(de),
                                        ctypes.byref(dptr), ctypes.byref(dlen), ctypes.byref(info),
                                        err, 2048)
    if rc == -2:
        return None
    if rc != 0:
        raise ValueError(err.value.decode())
    from genometools_amd import _lib
    lib = _lib.load()
    n = lib.gtamd_encoder_length(de)
    enc = np.empty(n, dtype=np.uint8)
    assert lib.gtamd_encoder_copy_symbols(de, enc.ctypes.data, 0, n) == 0
    desc = ctypes.string_at(dptr, dlen.value).split(b"\0")[:-1]
    fl = np.ctypeslib.as_array(ctypes.cast(info.filelengthtab, ctypes.POINTER(ctypes.c_uint64)),
                               shape=(2 * len(paths),)).copy().reshape(-1, 2)
    orig = list(info.originaldistribution)
    host.gtamd_encinfo_free(ctypes.byref(info))
    lib.gtamd_encoder_destroy(de)
    ctypes.CDLL(None).free(dptr)
    return enc, desc, orig, [tuple(int(x) for x in row) for row in fl]


def _fastq(rng, nrec, minlen, maxlen, alphabet="ACGTNacgtn", plusname=0.3):
    out = []
    for k in range(nrec):
        ln = int(rng.integers(minlen, maxlen + 1))
        name = b"r%d len=%d @x +y" % (k, ln) if k % 7 else b""
        seq = "".join(rng.choice(list(alphabet), size=ln)).encode()
        # qualities may start with '@' or '+' and contain any printable character
        qual = bytes(rng.integers(33, 127, size=ln, dtype=np.uint8))
        if k % 3 == 0:
            qual = b"@" + qual[1:]
        elif k % 3 == 1:
            qual = b"+" + qual[1:]
        out.append(b"@" + name + b"\n" + seq + b"\n+" + (name if rng.random() < plusname else b"") +
                   b"\n" + qual + b"\n")
    return b"".join(out)


FASTQ_STRICT = [[ou.fixture_path("test1.fastq")], [ou.fixture_path("fastq_long.fastq")],
                [os.path.join(ou.GOLDEN_DIR, "multi", "long_reads.fastq")],
                [os.path.join(ou.GOLDEN_DIR, "multi", f) for f in ("reads_a.fastq", "reads_b.fastq")],
                [os.path.join(ou.GOLDEN_DIR, "multi", f) for f in ("reads_b.fastq", "reads_a.fastq",
                                                                   "reads_b.fastq")]]


@pytest.mark.parametrize("paths", FASTQ_STRICT, ids=lambda p: "+".join(os.path.basename(x) for x in p))
def test_device_fastq_reader_matches_host_reader(gpu, host, paths):
    """symbols, descriptions, original characters and the file length table (booked
    per 8192-symbol buffer fill, src/core/sequence_buffer_fastq.c:42-191) of the
    reference's fixtures"""
    got = _device_encode_via_host(host, paths, False)
    assert got is not None
    want = _host_encode(host, paths, False)
    assert np.array_equal(got[0], want[0])
    assert got[1:] == want[1:]


def test_device_fastq_reader_generated_files(gpu, host, tmp_path):
    """records around the tile size of the kernels (4096 bytes) and the buffer size
    of the reference's reader (8192 symbols); '@' and '+' leading quality lines;
    empty names; repeated names on the '+' line; several files"""
    rng = np.random.default_rng(11)
    shapes = [(1, 1, 1), (5, 1, 3), (3, 4090, 4100), (40, 8185, 8200), (2000, 30, 250),
              (7, 20000, 30000), (300, 1, 2)]
    paths = []
    for i, (nrec, lo, hi) in enumerate(shapes):
        p = tmp_path / ("g%d.fastq" % i)
        p.write_bytes(_fastq(rng, nrec, lo, hi))
        paths.append(str(p))
        got = _device_encode_via_host(host, [str(p)], False)
        want = _host_encode(host, [str(p)], False)
        assert got is not None and np.array_equal(got[0], want[0]), shapes[i]
        assert got[1:] == want[1:], shapes[i]
    got = _device_encode_via_host(host, paths, False)
    want = _host_encode(host, paths, False)
    assert got is not None and np.array_equal(got[0], want[0]) and got[1:] == want[1:]
    # protein
    p = tmp_path / "p.fastq"
    p.write_bytes(_fastq(rng, 50, 10, 500, alphabet="ACDEFGHIKLMNPQRSTVWYXBZ"))
    got, want = _device_encode_via_host(host, [str(p)], True), _host_encode(host, [str(p)], True)
    assert got is not None and np.array_equal(got[0], want[0]) and got[1:] == want[1:]
    # the summary over FASTQ symbols, and the encoder's own interface
    with encode.DeviceEncoder() as de:
        de.encode(paths[:3])
        s, sym = de.summary(), de.symbols()
        file, seqlen, desclen = de.fastq_records()
        assert file.size == sum(sh[0] for sh in shapes[:3]) and s["numofsequences"] == file.size
        assert int(seqlen.sum()) + file.size - 1 == de.length
        assert [len(d) for d in de.descriptions()] == desclen.tolist()
    for k, v in _expected_summary(sym, 4).items():
        assert s[k] == v, k


def test_device_fastq_reader_declines_what_is_not_four_lines(gpu, host, tmp_path):
    """... and the host reader then reads it, or has the reference's message"""
    ok = b"@a\nACGT\n+\nIIII\n"
    cases = {
        "multiline": b"@a\nAC\nGT\n+\nII\nII\n",
        "no_last_newline": b"@a\nACGT\n+\nIIII",
        "crlf": b"@a\r\nACGT\r\n+\r\nIIII\r\n",
        "blank_in_sequence": b"@a\nAC GT\n+\nIIII\n",
        "blank_in_qualities": b"@a\nACGT\n+\nII II\n",
        "short_qualities": ok + b"@b\nACGT\n+\nIII\n",
        "long_qualities": b"@a\nACGT\n+\nIIIII\n" + ok,
        "other_name": b"@a\nACGT\n+b\nIIII\n",
        "illegal_symbol": ok + b"@b\nACXT\n+\nIIII\n",
        "empty_sequence": b"@a\n\n+\n\n",
        "no_at": ok + b"b\nACGT\n+\nIIII\n",
        "no_plus": b"@a\nACGT\nIIII\nIIII\n",
        "three_lines": ok + b"@b\nACGT\n+\n",
        "blank_line_between": ok + b"\n" + ok,
    }
    for name, raw in cases.items():
        p = tmp_path / (name + ".fastq")
        p.write_bytes(raw)
        assert _device_encode_via_host(host, [str(p)], False) is None, name
        with pytest.raises(encode.DeviceDeclined):
            encode.DeviceEncoder().encode([str(p)])
    # FASTA and FASTQ in one run stay with the host reader too
    q, a = tmp_path / "ok.fastq", tmp_path / "ok.fna"
    q.write_bytes(ok)
    a.write_bytes(b">x\nACGT\n")
    assert _device_encode_via_host(host, [str(q), str(a)], False) is None
    assert _device_encode_via_host(host, [str(q)], False) is not None
