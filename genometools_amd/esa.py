"""Host-side mirror of the reference's ESA construction interface over the C ABI.

The reference's seam is the Sfxiterator (src/match/sfx-suffixer.h:32-72):
``gt_Sfxiterator_new_withadditionalvalues`` / ``_next`` / ``_longest`` /
``_delete`` plus the LCP and BWT sinks of src/match/sfx-run.c.  `EsaEngine`
is the coarse-grained equivalent (one run, all tables resident on the device);
`Sfxiterator` below keeps the reference's call shape on top of it.

Everything here goes through genometools_amd/libgtamd_esa.so (HIP); there is
no CPU implementation in this package.
"""
import ctypes
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import EsaError, EsaStats, EsaTiming, check  # noqa: F401

WANT_SUF, WANT_LCP, WANT_BWT, WANT_BCK = 1, 2, 4, 8
TAB_SUF, TAB_LCP, TAB_BWT, TAB_LLV, TAB_BCK = 0, 1, 2, 3, 4

_TAB_DTYPE = {TAB_SUF: np.uint64, TAB_LCP: np.uint8, TAB_BWT: np.uint8,
              TAB_LLV: np.uint64, TAB_BCK: np.uint32}


@dataclass
class EsaResult:
    """the tables of one index, as the files .suf/.lcp/.llv/.bwt hold them"""
    suf: np.ndarray = None
    lcp: np.ndarray = None
    llv: np.ndarray = None
    bwt: np.ndarray = None
    stats: dict = None
    timing: dict = None


def _struct_dict(s):
    return {name: getattr(s, name) for name, _ in s._fields_}


class EsaEngine:
    """Device context: workspace for sequences of up to `max_n` symbols over
    an alphabet of `numofchars` letters (4 = DNA, 20 = protein)."""

    def __init__(self, max_n, numofchars=4, device=0):
        self._lib = _lib.load()
        self._ctx = self._lib.gtamd_esa_create(device, max_n, numofchars)
        if not self._ctx:
            raise EsaError(self._lib.gtamd_esa_last_error().decode())
        self.max_n = max_n
        self.numofchars = numofchars
        self.device = device
        self.n = None

    def close(self):
        if self._ctx:
            self._lib.gtamd_esa_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- input ------------------------------------------------------------
    def set_sequence(self, enc):
        """encoded symbols (numpy uint8: 0..sigma-1, 254, 255) from the host"""
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        check(self._lib.gtamd_esa_set_sequence_bytes(
            self._ctx, enc.ctypes.data_as(ctypes.c_void_p), enc.size, 0))
        self.n = int(enc.size)

    def set_sequence_device(self, ptr, n):
        """encoded symbols already resident on the device (raw pointer)"""
        check(self._lib.gtamd_esa_set_sequence_bytes(self._ctx, ptr, n, 1))
        self.n = int(n)

    def set_sequence_packed_device(self, twobit_ptr, specialbits_ptr, n):
        """already packed, device-resident input (GtTwobitencoding words +
        special bitmap, see include/gtamd_esa.h); read in place"""
        check(self._lib.gtamd_esa_set_sequence_packed(self._ctx, twobit_ptr,
                                                      specialbits_ptr, n))
        self.n = int(n)

    # -- sharding -----------------------------------------------------------
    def set_part(self, part, numparts, comm=None):
        """build slice `part` of `numparts` lexicographic ranges; `comm` is a
        genometools_amd.dist.TorchComm (kept alive by the engine)"""
        check(self._lib.gtamd_esa_set_part(self._ctx, part, numparts))
        self._comm = comm
        if comm is not None:
            check(self._lib.gtamd_esa_set_comm(
                self._ctx, ctypes.cast(comm.allgather_cb, ctypes.c_void_p),
                ctypes.cast(comm.alltoallv_cb, ctypes.c_void_p), None))

    def table_offset(self):
        return int(self._lib.gtamd_esa_table_offset(self._ctx))

    # -- hot path ---------------------------------------------------------
    def run(self, want=WANT_SUF | WANT_LCP | WANT_BWT):
        check(self._lib.gtamd_esa_run(self._ctx, want))
        self.want = want

    # -- output -----------------------------------------------------------
    def entries(self, which):
        return int(self._lib.gtamd_esa_table_entries(self._ctx, which))

    def device_pointer(self, which):
        return self._lib.gtamd_esa_table_device(self._ctx, which)

    def table(self, which, first=0, count=None):
        total = self.entries(which)
        count = total - first if count is None else count
        width = 2 if which == TAB_LLV else 1
        out = np.empty(count * width, dtype=_TAB_DTYPE[which])
        if count:
            check(self._lib.gtamd_esa_table_copy(
                self._ctx, which, out.ctypes.data_as(ctypes.c_void_p), first,
                count))
        return out.reshape(-1, 2) if which == TAB_LLV else out

    def set_readmode(self, readmode):
        """GtReadmode of the sequence handed in next (0 forward, 1 reverse,
        2 complement, 3 reverse complement; src/core/readmode.h)"""
        check(self._lib.gtamd_esa_set_readmode(self._ctx, readmode))

    def set_prefixlength(self, k):
        """prefix length of .prj and of the bucket table; 0: the recommended one"""
        check(self._lib.gtamd_esa_set_prefixlength(self._ctx, k))

    def bcktab(self):
        """the sections of INDEX.bck (run with WANT_BCK): leftborder,
        countspecialcodes, distpfxidx counters (src/match/bcktab.c:519-558)"""
        a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        check(self._lib.gtamd_esa_bck_layout(self._ctx, ctypes.byref(a), ctypes.byref(b),
                                             ctypes.byref(c)))
        raw = self.table(TAB_BCK)
        na, nb = a.value + 1, b.value
        return raw[:na], raw[na:na + nb], raw[na + nb:na + nb + c.value]

    def stats(self):
        s = EsaStats()
        check(self._lib.gtamd_esa_get_stats(self._ctx, ctypes.byref(s)))
        return _struct_dict(s)

    def timing(self):
        t = EsaTiming()
        check(self._lib.gtamd_esa_get_timing(self._ctx, ctypes.byref(t)))
        return _struct_dict(t)

    def result(self):
        r = EsaResult(stats=self.stats(), timing=self.timing())
        if self.want & WANT_SUF:
            r.suf = self.table(TAB_SUF)
        if self.want & WANT_LCP:
            r.lcp = self.table(TAB_LCP)
            r.llv = self.table(TAB_LLV)
        if self.want & WANT_BWT:
            r.bwt = self.table(TAB_BWT)
        return r


class Sfxiterator:
    """The reference's iterator seam on top of the one-shot engine
    (src/match/sfx-suffixer.h:32-72): `next()` hands out the suffix array in
    slices -- first all suffixes that start with a letter (what the reference
    delivers part by part), then the suffixes that start with a special in
    pages, `None` at the end (src/match/sfx-suffixer.c:2162-2198) -- and
    `longest()` is the index of suffix 0.  Arguments keep the reference's
    names; `readmode` is the reference's GtReadmode (0 fwd, 1 rev, 2 cpl,
    3 rcl), applied on the device while the sequence is packed;
    `numofparts`/`maximumspace` only set the page size of the special tail:
    the whole table is resident."""

    def __init__(self, encseq, readmode=0, prefixlength=0, numofparts=1,
                 maximumspace=0, numofchars=4, device=0):
        enc = np.ascontiguousarray(encseq, dtype=np.uint8)
        self._eng = EsaEngine(max(int(enc.size), 1), numofchars, device)
        check(self._eng._lib.gtamd_esa_set_prefixlength(self._eng._ctx, prefixlength))
        self._eng.set_readmode(readmode)
        self._eng.set_sequence(enc)
        self._eng.run(WANT_SUF)
        n1 = int(enc.size) + 1
        specials = int(np.count_nonzero(enc >= 254))
        self._nonspecial = n1 - specials - 1
        page = max(1, -(-self._nonspecial // max(1, numofparts)))
        self._slices = [(0, self._nonspecial, False)] if self._nonspecial else []
        first = self._nonspecial
        while first < n1:
            cnt = min(page, n1 - first)
            self._slices.append((first, cnt, True))
            first += cnt
        self._next = 0

    def next(self):
        """(suffixsortspace slice, numberofsuffixes, specialsuffixes) or None"""
        if self._next >= len(self._slices):
            return None
        first, cnt, special = self._slices[self._next]
        self._next += 1
        return self._eng.table(TAB_SUF, first, cnt), cnt, special

    def longest(self):
        return self._eng.stats()["longest"]

    def delete(self):
        self._eng.close()


def suffixerator_tables(enc, numofchars=4, want=WANT_SUF | WANT_LCP | WANT_BWT,
                        device=0):
    """one-shot: encoded symbols in, EsaResult out"""
    enc = np.ascontiguousarray(enc, dtype=np.uint8)
    with EsaEngine(max(int(enc.size), 1), numofchars, device) as eng:
        eng.set_sequence(enc)
        eng.run(want)
        return eng.result()


def pack_twobit(enc):
    """host-side packing of encoded DNA symbols into the resident form of the
    engine: GtTwobitencoding words (32 symbols per uint64, first symbol in the
    top bits, src/core/intbits.h:78-83; a special keeps its kind: 0 wildcard,
    1 separator) and the special bitmap with the virtual end bit set"""
    enc = np.ascontiguousarray(enc, dtype=np.uint8)
    n = int(enc.size)
    codes = np.where(enc >= 254, (enc == 255).astype(np.uint8), enc).astype(np.uint64)
    nw = max(1, -(-n // 32))
    padded = np.zeros(nw * 32, dtype=np.uint64)
    padded[:n] = codes
    shifts = (np.uint64(62) - np.uint64(2) * np.arange(32, dtype=np.uint64))
    twobit = np.bitwise_or.reduce(padded.reshape(nw, 32) << shifts, axis=1)
    nsw = -(-(n + 1) // 64)
    bits = np.zeros(nsw * 64, dtype=np.uint8)
    bits[:n] = enc >= 254
    bits[n] = 1
    special = np.packbits(bits.reshape(nsw, 64), axis=1, bitorder="little").view(np.uint64).reshape(nsw)
    return twobit, special


def prj_text(seqstats, esastats, with_lcp=True):
    """the .prj file of src/match/sfx-outprj.c:38-83 as text; `seqstats` are
    the encoded-sequence numbers (genometools_amd.encseq.sequence_stats),
    `esastats` the engine's"""
    n1 = esastats["numberofallsortedsuffixes"]
    lines = ["totallength=%d" % seqstats["totallength"]]
    for k in ("specialcharacters", "specialranges", "realspecialranges",
              "lengthofspecialprefix", "lengthofspecialsuffix", "wildcards",
              "wildcardranges", "realwildcardranges", "lengthofwildcardprefix",
              "lengthofwildcardsuffix"):
        lines.append("%s=%d" % (k, seqstats[k]))
    lines += ["numofsequences=%d" % seqstats["numofsequences"],
              "numofdbsequences=%d" % seqstats["numofsequences"],
              "numofquerysequences=0",
              "numberofallsortedsuffixes=%d" % n1,
              "longest=%d" % esastats["longest"],
              "prefixlength=%d" % esastats["prefixlength"],
              "largelcpvalues=%d" % (esastats["largelcpvalues"] if with_lcp else 0),
              "averagelcp=%.2f" % ((esastats["lcptabsum"] / n1) if with_lcp else 0.0),
              "maxbranchdepth=%d" % (esastats["maxbranchdepth"] if with_lcp else 0),
              "integersize=64", "littleendian=1", "readmode=0", "mirrored=0"]
    return "\n".join(lines) + "\n"
