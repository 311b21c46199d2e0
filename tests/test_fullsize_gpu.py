"""BASELINE.json sizes, checked through size-independent properties (the CPU
oracle's comparison sort cannot run at these sizes):

* configs[1], 256 Mbp uniform DNA: the suffix array passes the linear-time
  checker (permutation + sortedness, oracle/esa_oracle.c ora_check_suffix_array)
  and LCP/BWT equal the tables the oracle derives from that suffix array
  (Kasai).  Together that is bit-exactness of all three tables.
* configs[2], 3 Gbp human-like DNA (the bench workload): the suffix table
  EXACTLY -- permutation + every neighbour pair ordered by (first symbol, rank of
  the successor), the reference's lightweight check restated on the device
  (tests/device_check.py) --, .bwt for every entry, .lcp on 2 * 10^7 samples,
  every .llv entry, the tail layout (specials in text order, then n).
* configs[4], 10^9 protein residues: the same properties on the 5-bit path.
* the position range of configs[3] (n >= 2^32; the 24 Gbp input itself needs
  the 8 GPUs it is defined on): 2^32 + 4 M bases of uniform DNA built in two
  parts on the one GPU, 64-bit positions and ranks.
"""
import numpy as np
import pytest
import torch

import oracle_util as ou
from genometools_amd import _lib, esa, synth

pytestmark = pytest.mark.gpu


def _device_sequence(model, seed, n):
    lib = _lib.load()
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    _lib.check(lib.gtamd_synth_bytes(0, model, seed, n, buf.data_ptr()))
    torch.cuda.synchronize()
    return buf


class _Wrap:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr,
                                         "data": (ptr, False), "version": 2}


def test_config1_256mbp_uniform_bit_exact(gpu):
    n = 256 * 1000 * 1000
    buf = _device_sequence(synth.MODEL_UNIFORM_DNA, 42, n)
    with esa.EsaEngine(n, 4) as eng:
        eng.set_sequence_device(buf.data_ptr(), n)
        eng.run()
        res = eng.result()
    enc = buf.cpu().numpy()
    del buf
    rc, where = ou.check_suffix_array(enc, res.suf)
    assert rc == 0, (rc, where)
    t = ou.tables_given_sa(enc, res.suf)
    assert np.array_equal(res.lcp, t["lcp"])
    assert np.array_equal(res.llv, t["llv"])
    assert np.array_equal(res.bwt, t["bwt"])
    assert res.stats["maxbranchdepth"] == int(t["lcpfull"].max())
    assert res.stats["largelcpvalues"] == len(t["llv"])
    # the device generator and the numpy definition of the model agree
    assert np.array_equal(enc[:1 << 20], synth.generate(synth.MODEL_UNIFORM_DNA, 42, n, 0, 1 << 20))


def test_config2_3gbp_humanlike_exact(gpu):
    """BASELINE.json configs[2], the bench workload, with the switches the bench
    runs with (none).  The suffix table is checked EXACTLY, on the device
    (tests/device_check.py: the reference's lightweight check restated -- a
    permutation whose every neighbour pair is ordered by (first symbol, rank of the
    successor) is the sorted table); .bwt for every entry; .lcp on 2 * 10^7 sampled
    entries against the symbols; every one of the 117 M .llv entries (position,
    mismatch right behind its value, agreement at its last and at 16 random
    offsets); the statistics of .prj from the tables."""
    import device_check as dc
    n = 3 * 1000 * 1000 * 1000
    N = n + 1
    buf = _device_sequence(synth.MODEL_HUMANLIKE_DNA, 43, n)
    with esa.EsaEngine(n, 4) as eng:
        eng.set_sequence_device(buf.data_ptr(), n)
        eng.run()
        st = eng.stats()
        assert st["msd_big_entries"] > 0 and st["pair_suffixes"] > 3e8 and st["refine_rounds"] >= 9
        sa = dc.as_tensor(eng.device_pointer(esa.TAB_SUF), N, "<i8")
        lcp = dc.as_tensor(eng.device_pointer(esa.TAB_LCP), N, "|u1")
        bwt = dc.as_tensor(eng.device_pointer(esa.TAB_BWT), N, "|u1")
        ok, msg = dc.check_suffix_array_exact(sa, buf)
        assert ok, msg
        assert int(sa[st["longest"]].item()) == 0
        ok, msg = dc.check_bwt_exact(sa, buf, bwt)
        assert ok, msg
        # the special tail: every suffix that starts with a special, in text order, then n
        specials = int((buf >= 254).sum().item())
        tail = sa[N - 1 - specials:]
        assert int(tail[-1].item()) == n
        assert bool((buf[tail[:-1]] >= 254).all()) and bool((tail[1:] > tail[:-1]).all())
        assert int(buf[sa[N - 2 - specials]].item()) < 254
        assert bool((lcp[N - specials:] == 0).all())
        # .llv: all of it
        nl = eng.entries(esa.TAB_LLV)
        assert nl == st["largelcpvalues"] > 10 ** 8
        llv = dc.as_tensor(eng.device_pointer(esa.TAB_LLV), 2 * nl, "<i8").view(-1, 2)
        llv_idx, llv_val = llv[:, 0].contiguous(), llv[:, 1].contiguous()
        assert dc.count_lcp_overflows(lcp) == nl
        assert int(llv_val.max().item()) == st["maxbranchdepth"]
        ok, msg = dc.check_llv_all(sa, buf, lcp, llv_idx, llv_val)
        assert ok, msg
        # .lcp on samples (2 * 10^7 entries, a tenth of them where the LCP is large)
        g = torch.Generator(device="cuda:0")
        g.manual_seed(2026)
        idx = torch.randint(1, N - specials, (18_000_000,), device="cuda:0", generator=g)
        pick = llv_idx[torch.randint(0, nl, (2_000_000,), device="cuda:0", generator=g)]
        for part in (idx, pick, torch.clamp(pick + 1, max=N - 1)):
            for a in range(0, part.numel(), 1 << 22):
                ok, msg = dc.check_lcp_samples(sa, buf, lcp, llv_idx, llv_val, part[a:a + (1 << 22)])
                assert ok, msg


def _sampled_neighbours(eng, enc_of, n, lo, hi, rng, samples, index_offset=0, wildcard_ok=True):
    """order, LCP byte and BWT byte of `samples` neighbour pairs of the table
    slice held by `eng`, re-derived on the CPU; enc_of(a, b) delivers the
    encoded symbols [a, b)"""
    bad = 0
    for i in np.sort(rng.integers(lo, hi, samples)):
        i = int(i)
        p, q = (int(x) for x in eng.table(esa.TAB_SUF, i - 1, 2))
        lcpb = int(eng.table(esa.TAB_LCP, i, 1)[0])
        bwtb = int(eng.table(esa.TAB_BWT, i, 1)[0])
        span = 64
        while True:
            a, b = enc_of(p, min(n, p + span)), enc_of(q, min(n, q + span))
            m = min(len(a), len(b))
            neq = np.nonzero((a[:m] != b[:m]) | (a[:m] >= 254))[0]
            if len(neq) or m < span:
                l = int(neq[0]) if len(neq) else m
                break
            span *= 8
        ca = a[l] if l < len(a) else 255
        cb = b[l] if l < len(b) else 255
        ka = 256 + p + l if ca >= 254 else int(ca)
        kb = 256 + q + l if cb >= 254 else int(cb)
        bw = 254 if q == 0 else int(enc_of(q - 1, q)[0])
        bad += not (ka < kb and lcpb == min(l, 255) and bwtb == bw)
    return bad


def test_config4_protein_1g_properties(gpu):
    """BASELINE.json configs[4]: 10^9 residues over the 20-letter alphabet
    (src/core/alphabet.c:488-503), -suf -lcp (+ -bwt): the 5-bit symbol path at
    full size (7 sort passes, N/12 word indexing)"""
    n = 1000 * 1000 * 1000
    N = n + 1
    buf = _device_sequence(synth.MODEL_PROTEIN, 44, n)
    with esa.EsaEngine(n, 20) as eng:
        eng.set_sequence_device(buf.data_ptr(), n)
        eng.run()
        st = eng.stats()
        assert st["prefixlength"] == 5
        sa = torch.as_tensor(_Wrap(eng.device_pointer(esa.TAB_SUF), N, "<i8"), device="cuda:0")
        assert int(sa.sum().item()) == N * (N - 1) // 2
        sq = int((sa * sa).sum().item()) % (1 << 64)
        assert sq == ((N - 1) * N * (2 * N - 1) // 6) % (1 << 64)
        assert int(sa[st["longest"]].item()) == 0
        del sa
        enc = buf.cpu().numpy()
        del buf
        torch.cuda.empty_cache()
        specials = int(np.count_nonzero(enc >= 254))
        assert specials > 2_000_000          # ~3 M sequence borders + X
        # tail: separators and X in text order, then n
        tail = eng.table(esa.TAB_SUF, N - 1 - specials, specials + 1)
        assert tail[-1] == n
        assert np.all(enc[tail[:-1].astype(np.int64)] >= 254)
        assert np.all(np.diff(tail[:-1].astype(np.int64)) > 0)
        before_tail = eng.table(esa.TAB_SUF, N - 2 - specials, 1)[0]
        assert enc[int(before_tail)] < 254
        rng = np.random.default_rng(44)
        bad = _sampled_neighbours(eng, lambda a, b: enc[a:b], n, 1, N - specials - 1, rng, 4000)
        assert bad == 0
        # i.i.d. residues: no LCP near the byte limit
        assert st["largelcpvalues"] == 0 and st["maxbranchdepth"] < 64
        lcp = eng.table(esa.TAB_LCP)
        assert int(lcp.max()) == st["maxbranchdepth"]
        assert np.all(lcp[N - 1 - specials + 1:] == 0)
        # averagelcp of .prj: entries with >= prefixlength letters;
        # an upper bound here, the exact mask is checked at oracle sizes
        assert 0 < st["lcptabsum"] <= int(lcp.sum(dtype=np.uint64))


def test_positions_beyond_2p32_in_two_parts(gpu):
    """n just above 2^32 (the position range BASELINE.json configs[3], 24 Gbp
    on 8 GPUs, needs): two parts as threads on the one GPU, 64-bit positions
    and ranks, the slices of both parts checked on the device and by samples"""
    import threading
    from thread_comm import ThreadComm
    n = (1 << 32) + (1 << 22) + 12345
    N = n + 1
    seed = 45
    buf = _device_sequence(synth.MODEL_UNIFORM_DNA, seed, n)
    shared = ThreadComm(2, 0)
    engines = [esa.EsaEngine(n, 4) for _ in range(2)]
    errors = []

    def worker(r):
        try:
            eng = engines[r]
            eng.set_sequence_device(buf.data_ptr(), n)
            eng.set_part(r, 2, shared.view(r))
            eng.run()
        except Exception as e:   # noqa: BLE001
            errors.append((r, repr(e)))
            shared.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    try:
        assert not errors, errors
        del buf
        torch.cuda.empty_cache()

        def enc_of(a, b):
            return synth.generate(synth.MODEL_UNIFORM_DNA, seed, n, a, b)

        total, sq, top = 0, 0, 0
        offs = []
        rng = np.random.default_rng(7)
        for r, eng in enumerate(engines):
            cnt, off = eng.entries(esa.TAB_SUF), eng.table_offset()
            offs.append((off, cnt))
            st = eng.stats()
            sa = torch.as_tensor(_Wrap(eng.device_pointer(esa.TAB_SUF), cnt, "<i8"), device="cuda:0")
            total += int(sa.sum().item())            # (below 2^63 for each slice?  no: mod 2^64)
            sq += int((sa * sa).sum().item())
            top = max(top, int(sa.max().item()))
            del sa
            assert st["tied_suffixes"] > 0           # random 20-mers collide at this size
            assert _sampled_neighbours(eng, enc_of, n, 1, cnt - (1 if r == 1 else 0), rng, 1500) == 0
        assert offs[0][0] == 0 and offs[1][0] == offs[0][1] and offs[1][0] + offs[1][1] == N
        assert abs(offs[0][1] - offs[1][1]) < N // 50          # even ranges
        assert total % (1 << 64) == (N * (N - 1) // 2) % (1 << 64)
        assert sq % (1 << 64) == ((N - 1) * N * (2 * N - 1) // 6) % (1 << 64)
        assert top == n                               # the virtual end, a position >= 2^32
        # the border between the slices is in order too
        p = int(engines[0].table(esa.TAB_SUF, offs[0][1] - 1, 1)[0])
        q = int(engines[1].table(esa.TAB_SUF, 0, 1)[0])
        a, b = enc_of(p, min(n, p + 64)), enc_of(q, min(n, q + 64))
        m = min(len(a), len(b))
        d = int(np.nonzero(a[:m] != b[:m])[0][0])
        assert a[d] < b[d]
        assert int(engines[1].table(esa.TAB_LCP, 0, 1)[0]) == d
    finally:
        for eng in engines:
            eng.close()


# ---- the packed index at a size the oracle cannot reach --------------------------
def _bits(raw, pos, n):
    """n bits at bit position pos of a byte string, most significant first"""
    v = 0
    for b in range(pos, pos + n):
        v = (v << 1) | ((raw[b >> 3] >> (7 - (b & 7))) & 1)
    return v


def _unrank_block(comp, perm, sigma, B):
    """the block with composition index `comp` and permutation index `perm`
    (inverse of gt_block2IndexPair, src/match/eis-seqblocktranslate.c:436-540)"""
    from math import comb, factorial
    cnt, left = [], B
    for i in range(sigma - 1):
        k = sigma - i - 1
        v = 0
        while True:
            ways = comb(left - v + k - 1, k - 1)
            if comp < ways:
                break
            comp -= ways
            v += 1
        cnt.append(v)
        left -= v
    cnt.append(left)

    def arrangements(c):
        r = factorial(sum(c))
        for x in c:
            r //= factorial(x)
        return r
    out = []
    for _ in range(B):
        for s in range(sigma):
            if cnt[s] == 0:
                continue
            cnt[s] -= 1
            ways = arrangements(cnt)
            if perm < ways:
                out.append(s)
                break
            perm -= ways
            cnt[s] += 1
    return out, arrangements


def _check_packed_index(builder, N, get_bwt, get_suf, nspecial, sample, count_prefix_below=0):
    """header, sizes and sampled buckets of the image in `builder` (default options:
    block size 8, 8 blocks per bucket, locate interval 16, marks as counts) against
    the tables it was made from; get_bwt / get_suf(first, count) -> numpy"""
    import struct
    from math import comb
    inf = builder.info()
    B, K, L, sigma, locfreq = 8, 8, 64, 4, 16
    assert inf["num_buckets"] == (N + 1 + L - 1) // L
    header = builder.image(0, 8192).tobytes()
    assert header[:4] == b"BDX\0"
    voff, roff, seqlen = (struct.unpack_from("<Q", header, o)[0] for o in (28, 40, 52))
    assert (voff, roff, seqlen) == (inf["var_data_pos"], inf["range_enc_pos"], N)
    bits_ulong, vdob = struct.unpack_from("<I", header, 64)[0], struct.unpack_from("<I", header, 72)[0]
    assert bits_ulong == (N - 1).bit_length()
    cib = (comb(B + sigma - 1, sigma - 1) - 1).bit_length()
    cbb = struct.unpack_from("<I", header, 84 + 4 * sigma + 32 + 4)[0]
    cw_bits = sigma * bits_ulong + vdob + cbb + K * cib
    assert cw_bits == inf["cw_bits"]
    assert inf["file_bytes"] == roff + 8 + 16 * inf["num_regions"]
    assert roff == voff + (inf["var_bits"] + 7) // 8
    for j in sample:
        rb = (j * cw_bits) // 8
        cw = builder.image(inf["cw_data_pos"] + rb, cw_bits // 8 + 2).tobytes()
        at = j * cw_bits - rb * 8
        sums = [_bits(cw, at + s * bits_ulong, bits_ulong) for s in range(sigma)]
        var_off = _bits(cw, at + sigma * bits_ulong, vdob)
        pbits = _bits(cw, at + sigma * bits_ulong + vdob, cbb)
        if j * L <= count_prefix_below:
            before = get_bwt(0, j * L)
            assert sums == [int(np.count_nonzero(before == s)) for s in range(sigma)], j
        want = get_bwt(j * L, L)
        vb = var_off // 8
        var = builder.image(voff + vb, 1200).tobytes()
        vat = var_off - vb * 8
        used = 0
        for b in range(K):
            comp = _bits(cw, at + sigma * bits_ulong + vdob + cbb + b * cib, cib)
            block, arrangements = _unrank_block(comp, 0, sigma, B)
            ways = arrangements([block.count(s) for s in range(sigma)])
            pb = (ways - 1).bit_length() if ways > 1 else 0
            perm = _bits(var, vat + used, pb)
            used += pb
            block, _ = _unrank_block(comp, perm, sigma, B)
            w = want[b * B:(b + 1) * B]
            assert block == [int(x) if x < 254 else 0 for x in w], (j, b)
        assert used == pbits, j
        # locate marks (count mode): number, then (row in bucket, text position)
        suf = get_suf(j * L, L)
        nm = _bits(var, vat + used, 7)
        used += 7
        rows = np.arange(j * L, (j + 1) * L)
        marked = [i for i in range(L) if int(suf[i]) % locfreq == 0 or
                  bool(want[i] >= 254) != bool(rows[i] >= N - nspecial)]
        assert nm == len(marked), j
        for i in marked:
            assert _bits(var, vat + used, 6) == i
            assert _bits(var, vat + used + 6, bits_ulong) == int(suf[i]), (j, i)
            used += 6 + bits_ulong
    return inf, roff


def test_packed_index_of_a_1gbp_sequence_decodes_to_the_bwt(gpu):
    """INDEX.bdx of 10^9 bases (human-like model: wildcard runs, separators), built
    from the resident tables: sampled buckets decode -- occurrence counters,
    composition and permutation index of every block, locate marks -- to the .bwt
    and .suf tables the image was made from; the region list is the list of the
    runs of specials in the BWT; sizes add up"""
    import struct
    from genometools_amd import pck
    n = 1000 * 1000 * 1000
    buf = _device_sequence(synth.MODEL_HUMANLIKE_DNA, 43, n)
    with esa.EsaEngine(n, 4) as eng, pck.PackedIndex() as builder:
        eng.set_sequence_device(buf.data_ptr(), n)
        del buf
        eng.run(esa.WANT_SUF | esa.WANT_BWT)
        builder.build_from_esa(eng)
        bwt = eng.table(esa.TAB_BWT)
        N = n + 1
        special = bwt >= 254
        rng = np.random.default_rng(9)
        nb = (N + 1 + 63) // 64
        sample = sorted(set(int(x) for x in rng.integers(0, nb - 1, 300)) |
                        {0, 1, 100, 5000, 200000, nb - 2})
        inf, roff = _check_packed_index(
            builder, N, lambda first, count: bwt[first:first + count],
            lambda first, count: eng.table(esa.TAB_SUF, first, count),
            int(np.count_nonzero(special)), sample, count_prefix_below=5 * 10 ** 7)
        # region list == runs of specials in the BWT
        change = np.flatnonzero(np.diff(bwt.astype(np.int16)) != 0) + 1
        starts = np.concatenate(([0], change))
        starts = starts[special[starts]]
        regions = builder.image(roff, 8 + 16 * inf["num_regions"]).tobytes()
        assert struct.unpack_from("<Q", regions, 0)[0] == inf["num_regions"] == starts.size + 1
        rec = np.frombuffer(regions, dtype=np.uint64, offset=8).reshape(-1, 2)
        assert np.array_equal(rec[:-1, 0], starts.astype(np.uint64))
        assert int(rec[-1, 0]) == N + 8


def test_packed_index_beyond_2p32_positions(gpu):
    """more than 2^32 table entries (33-bit counters and text positions, var offsets
    beyond 2^32 bits): tables made up on the device -- the builder takes any .bwt /
    .suf pair -- and sampled buckets of the image decoded against them"""
    from genometools_amd import pck
    N = (1 << 32) + (1 << 20) + 7
    dev = "cuda:0"
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    bwt = torch.empty(N, dtype=torch.uint8, device=dev)
    suf = torch.empty(N, dtype=torch.int64, device=dev)
    step = 1 << 28
    for o in range(0, N, step):
        m = min(step, N - o)
        x = torch.randint(0, 4, (m,), dtype=torch.uint8, device=dev, generator=g)
        u = torch.rand(m, device=dev, generator=g)
        x[u < 0.01] = 254
        bwt[o:o + m] = x
        i = torch.arange(o, o + m, dtype=torch.int64, device=dev)
        suf[o:o + m] = (i * 11400714819 + 12345) % N          # any values below N do
        del x, u, i
    # the tail of the table holds the suffixes that start with a special: as many
    # rows as the BWT holds specials
    nspecial = int((bwt >= 254).sum().item())
    with pck.PackedIndex() as builder:
        builder.build(bwt.data_ptr(), suf.data_ptr(), N, 4, 5)
        rng = np.random.default_rng(3)
        nb = (N + 1 + 63) // 64
        sample = sorted(set(int(x) for x in rng.integers(0, nb - 1, 200)) |
                        {0, 1, (1 << 26) - 1, 1 << 26, (1 << 26) + 1, nb - 2})
        inf, _ = _check_packed_index(
            builder, N, lambda first, count: bwt[first:first + count].cpu().numpy(),
            lambda first, count: suf[first:first + count].cpu().numpy().astype(np.uint64),
            nspecial, sample, count_prefix_below=10 ** 7)
        assert inf["var_bits"] > 1 << 32
