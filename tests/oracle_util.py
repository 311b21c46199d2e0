"""ctypes access to the CPU oracle (oracle/_build/libesa_oracle.so) and to the
golden vectors.  Test infrastructure only."""
import ctypes
import gzip
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "_build", "libesa_oracle.so")
ORACLE_CLI = os.path.join(ORACLE_DIR, "_build", "esa_oracle")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "gt_ref_sfx")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


class SeqStats(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint64) for k in (
        "totallength", "specialcharacters", "specialranges",
        "realspecialranges", "lengthofspecialprefix", "lengthofspecialsuffix",
        "wildcards", "wildcardranges", "realwildcardranges",
        "lengthofwildcardprefix", "lengthofwildcardsuffix",
        "numofsequences")] + [("numofchars", ctypes.c_uint32)]


class EsaStats(ctypes.Structure):
    _fields_ = [("numberofallsortedsuffixes", ctypes.c_uint64),
                ("longest", ctypes.c_uint64),
                ("largelcpvalues", ctypes.c_uint64),
                ("maxbranchdepth", ctypes.c_uint64),
                ("lcptabsum", ctypes.c_double),
                ("prefixlength", ctypes.c_uint32)]


class PckParams(ctypes.Structure):
    _fields_ = [("block_size", ctypes.c_uint), ("bucket_blocks", ctypes.c_uint),
                ("locate_interval", ctypes.c_uint), ("feature_toggles", ctypes.c_int),
                ("with_statistics", ctypes.c_int)]


_lib = None


def build():
    src = [os.path.join(ORACLE_DIR, f) for f in
           ("esa_oracle.c", "esa_oracle.h", "esa_oracle_main.c", "pck_oracle.c",
            "pck_oracle.h")]
    if (not os.path.exists(ORACLE_LIB) or not os.path.exists(ORACLE_CLI) or
            any(os.path.getmtime(s) > os.path.getmtime(ORACLE_LIB) for s in src)):
        subprocess.run(["make", "-C", ORACLE_DIR], check=True,
                       stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(ORACLE_LIB)
        P, U64, U32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32
        L.ora_encode_fasta.argtypes = [ctypes.c_char_p, ctypes.c_int,
                                       ctypes.POINTER(P), ctypes.POINTER(U64),
                                       ctypes.c_char_p, ctypes.c_size_t]
        L.ora_seqstats_compute.argtypes = [P, U64, U32, U64, U64,
                                           ctypes.POINTER(SeqStats)]
        L.ora_recommended_prefixlength.restype = U32
        L.ora_recommended_prefixlength.argtypes = [U32, U64]
        L.ora_suffix_array.argtypes = [P, U64, P]
        L.ora_lcp_direct.argtypes = [P, U64, P, P]
        L.ora_lcp_kasai.argtypes = [P, U64, P, P]
        L.ora_bwt.argtypes = [P, U64, P, P]
        L.ora_lcp_to_bytes.restype = U64
        L.ora_lcp_to_bytes.argtypes = [P, U64, P, P]
        L.ora_esastats_compute.argtypes = [P, U64, P, P, U32,
                                           ctypes.POINTER(EsaStats)]
        L.ora_check_suffix_array.argtypes = [P, U64, P, ctypes.POINTER(U64)]
        L.ora_pck_default_toggles.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint,
                                              ctypes.c_int]
        L.ora_pck_bdx.argtypes = [P, P, P, U64, ctypes.c_uint, U64, ctypes.POINTER(PckParams),
                                  ctypes.POINTER(P), ctypes.POINTER(ctypes.c_size_t)]
        L.ora_pck_free.argtypes = [P]
        L.ora_pck_last_var_bits.restype = U64
        L.ora_pck_ctxmap.argtypes = [P, U64, ctypes.c_int, ctypes.POINTER(P),
                                     ctypes.POINTER(ctypes.c_size_t)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def encode_fasta(path, protein=False):
    L = lib()
    ptr, n = ctypes.c_void_p(), ctypes.c_uint64()
    err = ctypes.create_string_buffer(1024)
    rc = L.ora_encode_fasta(path.encode(), int(protein), ctypes.byref(ptr),
                            ctypes.byref(n), err, 1024)
    if rc != 0:
        raise ValueError(err.value.decode())
    enc = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)),
                                shape=(n.value,)).copy()
    ctypes.CDLL(None).free(ptr)
    return enc


def esa(enc, numofchars=4, kasai=True):
    """all tables + statistics of the oracle for encoded symbols `enc`"""
    L = lib()
    enc = np.ascontiguousarray(enc, dtype=np.uint8)
    n = enc.size
    sa = np.empty(n + 1, dtype=np.uint64)
    L.ora_suffix_array(_p(enc), n, _p(sa))
    lcpw = np.empty(n + 1, dtype=np.uint64)
    (L.ora_lcp_kasai if kasai else L.ora_lcp_direct)(_p(enc), n, _p(sa), _p(lcpw))
    lcpb = np.empty(n + 1, dtype=np.uint8)
    pairs = L.ora_lcp_to_bytes(_p(lcpw), n + 1, _p(lcpb), None)
    llv = np.empty(2 * pairs, dtype=np.uint64)
    L.ora_lcp_to_bytes(_p(lcpw), n + 1, _p(lcpb), _p(llv))
    bwt = np.empty(n + 1, dtype=np.uint8)
    L.ora_bwt(_p(enc), n, _p(sa), _p(bwt))
    st = EsaStats()
    k = L.ora_recommended_prefixlength(numofchars, n)
    L.ora_esastats_compute(_p(enc), n, _p(sa), _p(lcpw), k, ctypes.byref(st))
    stats = {name: getattr(st, name) for name, _ in st._fields_}
    return {"suf": sa, "lcp": lcpb, "llv": llv.reshape(-1, 2), "bwt": bwt,
            "lcpfull": lcpw, "stats": stats}


def bck_sizes(numofchars, k):
    """entries of the three sections of INDEX.bck (src/match/bcktab.c:240-287)"""
    codes = numofchars ** k
    special = numofchars ** (k - 1) if k >= 1 else 1
    dist = sum(numofchars ** i for i in range(1, k - 1))
    return codes, special, dist


def bck_file_bytes(sections):
    """uint32 sections -> file content: every section padded to 8 bytes
    (src/core/mapspec.c:350-457)"""
    out = b""
    for sec in sections:
        raw = np.ascontiguousarray(sec, dtype="<u4").tobytes()
        if raw:
            out += raw + b"\0" * (-len(raw) % 8)
    return out


def bcktab(enc, numofchars, k):
    """the oracle's bucket table sections for prefix length k"""
    L = lib()
    enc = np.ascontiguousarray(enc, dtype=np.uint8)
    codes, special, dist = bck_sizes(numofchars, k)
    lb = np.zeros(codes + 1, dtype=np.uint32)
    cs = np.zeros(special, dtype=np.uint32)
    dp = np.zeros(max(dist, 1), dtype=np.uint32)
    L.ora_bcktab(_p(enc), enc.size, numofchars, k, _p(lb), _p(cs), _p(dp))
    return lb, cs, dp[:dist]


def tables_given_sa(enc, sa):
    """LCP (Kasai) and BWT of the oracle for an externally supplied suffix
    array -- used at sizes where the oracle's comparison sort is too slow"""
    L = lib()
    enc = np.ascontiguousarray(enc, dtype=np.uint8)
    sa = np.ascontiguousarray(sa, dtype=np.uint64)
    n = enc.size
    lcpw = np.empty(n + 1, dtype=np.uint64)
    L.ora_lcp_kasai(_p(enc), n, _p(sa), _p(lcpw))
    lcpb = np.empty(n + 1, dtype=np.uint8)
    pairs = L.ora_lcp_to_bytes(_p(lcpw), n + 1, _p(lcpb), None)
    llv = np.empty(2 * pairs, dtype=np.uint64)
    L.ora_lcp_to_bytes(_p(lcpw), n + 1, _p(lcpb), _p(llv))
    bwt = np.empty(n + 1, dtype=np.uint8)
    L.ora_bwt(_p(enc), n, _p(sa), _p(bwt))
    return {"lcp": lcpb, "llv": llv.reshape(-1, 2), "bwt": bwt, "lcpfull": lcpw}


def check_suffix_array(enc, sa):
    """0 if `sa` is the suffix array of `enc` under the reference's ordering"""
    L = lib()
    enc = np.ascontiguousarray(enc, dtype=np.uint8)
    sa = np.ascontiguousarray(sa, dtype=np.uint64)
    where = ctypes.c_uint64()
    rc = L.ora_check_suffix_array(_p(enc), enc.size, _p(sa), ctypes.byref(where))
    return rc, where.value


def seqstats(enc, numofchars=4):
    L = lib()
    enc = np.ascontiguousarray(enc, dtype=np.uint8)
    st = SeqStats()
    L.ora_seqstats_compute(_p(enc), enc.size, numofchars, 0, 1, ctypes.byref(st))
    return {name: getattr(st, name) for name, _ in st._fields_}


def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


def golden_table(name, ext):
    path = os.path.join(GOLDEN_DIR, "tables", "%s.%s.gz" % (name, ext))
    if not os.path.exists(path):
        return None
    with gzip.open(path, "rb") as f:
        raw = f.read()
    return np.frombuffer(raw, dtype=np.uint64 if ext in ("suf", "llv") else np.uint8)


def fixture_path(name):
    if name.startswith("extra/"):
        return os.path.join(GOLDEN_DIR, name)
    return os.path.join(GOLDEN_DIR, "fixtures", name)


def pck_default_toggles(bsize=8, blbuck=8, locfreq=16, locbitmap=None, sprank=False):
    """feature toggles `gt packedindex` derives from its options"""
    return lib().ora_pck_default_toggles(bsize, blbuck, locfreq,
                                         -1 if locbitmap is None else int(locbitmap)) | \
        (4 if sprank else 0)


def pck_bdx(enc, numofchars, suf, bwt, bsize=8, blbuck=8, locfreq=16, locbitmap=None,
            mkindex=False, sprank=False):
    """bytes of INDEX.bdx by the oracle's restatement: as `gt packedindex
    trsuftab` writes it, or (mkindex) as `gt packedindex mkindex` does"""
    L = lib()
    enc = np.ascontiguousarray(enc, dtype=np.uint8)
    suf = np.ascontiguousarray(suf, dtype=np.uint64)
    bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
    longest = int(np.flatnonzero(suf == 0)[0])
    pp = PckParams(bsize, blbuck, locfreq,
                   pck_default_toggles(bsize, blbuck, locfreq, locbitmap, sprank), int(mkindex))
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    rc = L.ora_pck_bdx(_p(bwt), _p(suf), _p(enc), enc.size + 1, numofchars, longest,
                       ctypes.byref(pp), ctypes.byref(out), ctypes.byref(n))
    if rc != 0:
        raise ValueError("ora_pck_bdx: %d" % rc)
    raw = ctypes.string_at(out, n.value)
    L.ora_pck_free(out)
    return raw


def pck_ctxmap(suf, ilog):
    """(interval log used, bytes of INDEX.<ilog>cxm) by the oracle's restatement"""
    L = lib()
    suf = np.ascontiguousarray(suf, dtype=np.uint64)
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    used = L.ora_pck_ctxmap(_p(suf), suf.size, ilog, ctypes.byref(out), ctypes.byref(n))
    if used < 0:
        raise ValueError("invalid context map interval %d" % ilog)
    raw = ctypes.string_at(out, n.value)
    L.ora_pck_free(out)
    return used, raw


def golden_ctxmap():
    with open(os.path.join(GOLDEN_DIR, "golden_ctxmap.json")) as f:
        return json.load(f)


def golden_pck():
    with open(os.path.join(GOLDEN_DIR, "golden_pck.json")) as f:
        return json.load(f)


def parse_pck_key(key):
    """'name|bsize=..|blbuck=..|locfreq=..|locbitmap=auto' -> (name, kwargs)"""
    parts = key.split("|")
    kw = dict(p.split("=") for p in parts[1:])
    out = dict(bsize=int(kw["bsize"]), blbuck=int(kw["blbuck"]), locfreq=int(kw["locfreq"]),
               locbitmap={"auto": None, "yes": True, "no": False}[kw["locbitmap"]])
    if kw.get("mode") == "mkindex":
        out["mkindex"] = True
    if kw.get("sprank") == "yes":
        out["sprank"] = True
    if "dir" in kw:
        out["direction"] = kw["dir"]
    return parts[0], out


def apply_readmode(enc, direction):
    """the sequence as the reference reads it with -dir fwd|rev|cpl|rcl
    (src/core/readmode_api.h:24-27; complement for DNA letters only)"""
    enc = np.ascontiguousarray(enc, dtype=np.uint8).copy()
    mode = {"fwd": 0, "rev": 1, "cpl": 2, "rcl": 3}[direction]
    L = lib()
    L.ora_apply_readmode.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int]
    L.ora_apply_readmode(_p(enc), enc.size, mode)
    return enc
