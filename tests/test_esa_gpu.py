"""Parity of the HIP path (through the C ABI) with the reference's golden
tables and with the CPU oracle on the same inputs.  Bit-exact: all tables are
integer/byte/index data."""
import hashlib

import numpy as np
import pytest

import oracle_util as ou
from genometools_amd import esa, synth

pytestmark = pytest.mark.gpu
GOLDEN = ou.golden()


def _md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


def _prj_value(prj, key):
    for line in prj.splitlines():
        if line.startswith(key + "="):
            return line.split("=", 1)[1]
    raise KeyError(key)


def _assert_same_as_oracle(enc, sigma, res, ora=None):
    ora = ou.esa(enc, sigma) if ora is None else ora
    assert np.array_equal(res.suf, ora["suf"]), "suf"
    assert np.array_equal(res.bwt, ora["bwt"]), "bwt"
    assert np.array_equal(res.lcp, ora["lcp"]), "lcp"
    assert np.array_equal(res.llv, ora["llv"]), "llv"
    st = ora["stats"]
    assert res.stats["longest"] == st["longest"]
    assert res.stats["largelcpvalues"] == st["largelcpvalues"]
    assert res.stats["maxbranchdepth"] == st["maxbranchdepth"]
    assert res.stats["lcptabsum"] == int(st["lcptabsum"])
    assert res.stats["prefixlength"] == st["prefixlength"]


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_reference_fixtures(gpu, name):
    """every suffixerator fixture of the reference's test suite
    (testsuite/gt_suffixerator_include.rb:119-143,290-308)"""
    e = GOLDEN[name]
    protein = e["alphabet"] == "protein"
    enc = ou.encode_fasta(ou.fixture_path(name), protein)
    res = esa.suffixerator_tables(enc, 20 if protein else 4)
    assert _md5(res.suf) == e["tables"]["suf"]["md5"]
    assert _md5(res.lcp) == e["tables"]["lcp"]["md5"]
    assert _md5(res.llv) == e["tables"]["llv"]["md5"]
    assert _md5(res.bwt) == e["tables"]["bwt"]["md5"]
    prj = e["prj"]
    n1 = res.stats["numberofallsortedsuffixes"]
    assert str(n1) == _prj_value(prj, "numberofallsortedsuffixes")
    assert str(res.stats["longest"]) == _prj_value(prj, "longest")
    assert str(res.stats["prefixlength"]) == _prj_value(prj, "prefixlength")
    assert str(res.stats["largelcpvalues"]) == _prj_value(prj, "largelcpvalues")
    assert str(res.stats["maxbranchdepth"]) == _prj_value(prj, "maxbranchdepth")
    assert "%.2f" % (res.stats["lcptabsum"] / n1) == _prj_value(prj, "averagelcp")
    ss = ou.seqstats(enc, 20 if protein else 4)
    assert esa.prj_text(ss, res.stats) == prj


@pytest.mark.parametrize("n", [0, 1, 2, 3, 27, 28, 29, 31, 32, 33, 63, 64, 65,
                               255, 256, 1000, 4095, 4096, 4097, 20000])
def test_uniform_dna_small(gpu, n):
    enc = synth.generate(synth.MODEL_UNIFORM_DNA, 42, n)
    res = esa.suffixerator_tables(enc, 4)
    _assert_same_as_oracle(enc, 4, res)


@pytest.mark.parametrize("model,sigma,n,seed", [
    (synth.MODEL_UNIFORM_DNA, 4, 1 << 20, 1),
    (synth.MODEL_HUMANLIKE_DNA, 4, 70000, 2),
    (synth.MODEL_HUMANLIKE_DNA, 4, 600000, 3),
    (synth.MODEL_PROTEIN, 20, 50000, 4),
    (synth.MODEL_PROTEIN, 20, 400000, 5),
    (synth.MODEL_REPEAT_HEAVY, 4, 300000, 6),     # half the blocks copies, satellite arrays
])
def test_synthetic_models(gpu, model, sigma, n, seed):
    enc = synth.generate(model, seed, n)
    res = esa.suffixerator_tables(enc, sigma)
    _assert_same_as_oracle(enc, sigma, res)


def _cases():
    rng = np.random.default_rng(7)
    a = rng.integers(0, 4, 3000, dtype=np.uint8)
    yield "all_same_letter", np.zeros(5000, dtype=np.uint8)
    yield "all_T", np.full(3000, 3, dtype=np.uint8)
    yield "all_wildcards", np.full(777, 254, dtype=np.uint8)
    yield "all_separators", np.full(130, 255, dtype=np.uint8)
    yield "period_2", np.tile(np.array([0, 1], dtype=np.uint8), 4000)
    yield "period_7", np.tile(np.array([0, 1, 1, 2, 3, 0, 2], dtype=np.uint8), 1500)
    yield "two_identical_sequences", np.concatenate([a, [255], a]).astype(np.uint8)
    yield "identical_then_wildcard", np.concatenate([a, [254], a, [254]]).astype(np.uint8)
    yield "special_prefix_and_suffix", np.concatenate(
        [[254] * 40, a[:500], [255], [254] * 3, a[:500], [254] * 70]).astype(np.uint8)
    yield "T_runs_before_specials", np.concatenate(
        [[3] * 40, [254], [3] * 30, [255], [3] * 28, [254], [3] * 27, [254],
         [3] * 29]).astype(np.uint8)
    yield "specials_every_other", np.tile(np.array([2, 254], dtype=np.uint8), 500)
    # long exact repeat far beyond 255: .llv must carry it
    b = rng.integers(0, 4, 20000, dtype=np.uint8)
    yield "long_repeat", np.concatenate([b, a[:10], b, a[:7], b[:5000]]).astype(np.uint8)


@pytest.mark.parametrize("name,enc", list(_cases()), ids=[c[0] for c in _cases()])
def test_edge_cases(gpu, name, enc):
    res = esa.suffixerator_tables(enc, 4)
    _assert_same_as_oracle(enc, 4, res)


def test_protein_edge_cases(gpu):
    rng = np.random.default_rng(11)
    a = rng.integers(0, 20, 2000, dtype=np.uint8)
    for enc in (np.full(300, 19, dtype=np.uint8),
                np.concatenate([a, [255], a, [254], a[:100]]).astype(np.uint8),
                np.tile(np.arange(13, dtype=np.uint8), 200)):
        res = esa.suffixerator_tables(enc, 20)
        _assert_same_as_oracle(enc, 20, res)


def test_want_subsets_and_reuse(gpu):
    """-suf only, -lcp only, -bwt only give the same tables; a context can be
    reused for several sequences"""
    enc1 = synth.generate(synth.MODEL_HUMANLIKE_DNA, 9, 200000)
    enc2 = synth.generate(synth.MODEL_UNIFORM_DNA, 10, 50000)
    with esa.EsaEngine(200000, 4) as eng:
        for enc in (enc1, enc2, enc1):
            ora = ou.esa(enc, 4)
            eng.set_sequence(enc)
            for want in (esa.WANT_SUF, esa.WANT_LCP, esa.WANT_BWT,
                         esa.WANT_SUF | esa.WANT_LCP | esa.WANT_BWT):
                eng.run(want)
                r = eng.result()
                if want & esa.WANT_SUF:
                    assert np.array_equal(r.suf, ora["suf"])
                if want & esa.WANT_LCP:
                    assert np.array_equal(r.lcp, ora["lcp"])
                    assert np.array_equal(r.llv, ora["llv"])
                if want & esa.WANT_BWT:
                    assert np.array_equal(r.bwt, ora["bwt"])


def test_large_uniform_properties(gpu):
    """16 Mbp: the oracle's comparison sort is too slow here, so check the
    size-independent properties: the linear-time suffix-array checker
    (sortedness + permutation) and LCP/BWT recomputed by Kasai from the
    engine's own suffix array"""
    n = 16 * 1000 * 1000
    enc = synth.generate(synth.MODEL_UNIFORM_DNA, 42, n)
    res = esa.suffixerator_tables(enc, 4)
    rc, where = ou.check_suffix_array(enc, res.suf)
    assert rc == 0, (rc, where)
    t = ou.tables_given_sa(enc, res.suf)
    assert np.array_equal(res.lcp, t["lcp"])
    assert np.array_equal(res.llv, t["llv"])
    assert np.array_equal(res.bwt, t["bwt"])
    assert res.stats["maxbranchdepth"] == int(t["lcpfull"].max())


def test_errors(gpu):
    with pytest.raises(esa.EsaError, match="exceeds the context capacity"):
        with esa.EsaEngine(100, 4) as eng:
            eng.set_sequence(np.zeros(101, dtype=np.uint8))
    with pytest.raises(esa.EsaError, match="no sequence set"):
        with esa.EsaEngine(100, 4) as eng:
            eng.run()
    with pytest.raises(esa.EsaError, match="exceeds the position range"):
        esa.EsaEngine(1 << 41, 4)
    # beyond 2^32 positions only a part build can run (32-bit indices per slice)
    with esa.EsaEngine(1 << 33, 4) as eng:
        eng.set_sequence(np.zeros(10, dtype=np.uint8))
        eng.run()
    with pytest.raises(esa.EsaError, match="only defined for DNA"):
        with esa.EsaEngine(100, 20) as eng:
            eng.set_readmode(2)


def test_sfxiterator_adapter(gpu):
    """the reference's iterator call shape: slices concatenate to .suf, the
    special tail comes in pages flagged `specialsuffixes`"""
    e = GOLDEN["Atinsert.fna"]
    enc = ou.encode_fasta(ou.fixture_path("Atinsert.fna"))
    sfi = esa.Sfxiterator(enc, readmode=0, prefixlength=0, numofparts=3)
    parts, flags = [], []
    while True:
        r = sfi.next()
        if r is None:
            break
        tab, cnt, special = r
        assert len(tab) == cnt
        parts.append(tab)
        flags.append(special)
    assert flags[0] is False and all(flags[1:])
    assert len(parts[0]) == 11817 + 1 - 2950 - 1       # non-special suffixes
    assert _md5(np.concatenate(parts)) == e["tables"]["suf"]["md5"]
    assert sfi.longest() == 2529
    sfi.delete()


def _apply_readmode(enc, mode):
    out = enc[::-1].copy() if mode & 1 else enc.copy()
    if mode & 2:
        out = np.where(out < 4, 3 - out, out).astype(np.uint8)
    return out


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_readmode_at_the_library_seam(gpu, mode):
    """the `readmode` argument of gt_Sfxiterator_new_withadditionalvalues
    (src/match/sfx-suffixer.h:48-60): reverse / complement applied on the device
    while the sequence is packed; the tables are those of the transformed
    sequence (tests/test_cli_gpu.py holds the reference's files for -dir)"""
    enc = synth.generate(synth.MODEL_HUMANLIKE_DNA, 31, 150001)
    enc[:7] = 254          # the special prefix becomes a special suffix
    ora = ou.esa(_apply_readmode(enc, mode), 4)
    with esa.EsaEngine(enc.size, 4) as eng:
        eng.set_readmode(mode)
        eng.set_sequence(enc)
        eng.run()
        _assert_same_as_oracle(None, 4, eng.result(), ora)
        eng.set_readmode(0)
        eng.set_sequence(enc)
        eng.run()
        _assert_same_as_oracle(enc, 4, eng.result())
    sfi = esa.Sfxiterator(enc, readmode=mode)
    parts = []
    while True:
        r = sfi.next()
        if r is None:
            break
        parts.append(r[0])
    assert np.array_equal(np.concatenate(parts), ora["suf"])
    assert sfi.longest() == ora["stats"]["longest"]
    sfi.delete()
    if mode == 1:   # protein: reverse is defined, complement is not
        p = synth.generate(synth.MODEL_PROTEIN, 8, 30000)
        with esa.EsaEngine(p.size, 20) as eng:
            eng.set_readmode(1)
            eng.set_sequence(p)
            eng.run()
            _assert_same_as_oracle(None, 20, eng.result(), ou.esa(p[::-1].copy(), 20))


def test_packed_device_input(gpu):
    """gtamd_esa_set_sequence_packed: the caller's own GtTwobitencoding words
    and special bitmap, resident on the device, read in place"""
    import torch
    for enc in (synth.generate(synth.MODEL_HUMANLIKE_DNA, 21, 300000),
                synth.generate(synth.MODEL_UNIFORM_DNA, 22, 64),
                synth.generate(synth.MODEL_UNIFORM_DNA, 23, 1000)[:31],
                np.concatenate([[254, 255], synth.generate(0, 24, 500), [254] * 70]).astype(np.uint8)):
        twobit, special = esa.pack_twobit(enc)
        d_tb = torch.from_numpy(twobit.view(np.int64)).to("cuda:0")
        d_sp = torch.from_numpy(special.view(np.int64)).to("cuda:0")
        with esa.EsaEngine(enc.size, 4) as eng:
            eng.set_sequence_packed_device(d_tb.data_ptr(), d_sp.data_ptr(), enc.size)
            eng.run()
            res = eng.result()
        _assert_same_as_oracle(enc, 4, res)
    with pytest.raises(esa.EsaError, match="2-bit DNA layout only"):
        with esa.EsaEngine(100, 20) as eng:
            eng.set_sequence_packed_device(d_tb.data_ptr(), d_sp.data_ptr(), 10)


@pytest.mark.parametrize("model,n,ks", [
    (synth.MODEL_HUMANLIKE_DNA, 1_000_003, [0, 1, 2, 5, 12]),
    (synth.MODEL_UNIFORM_DNA, 300_000, [0, 3, 9]),
    (synth.MODEL_PROTEIN, 400_000, [0, 1, 2, 3, 4]),
])
def test_bucket_table_matches_oracle(gpu, model, n, ks):
    """GTAMD_WANT_BCK: the three sections of INDEX.bck from the sorted keys,
    against the oracle's restatement (pinned to the reference's .bck files in
    tests/test_oracle_golden.py)"""
    sigma = synth.numofchars(model)
    enc = synth.generate(model, 11, n)
    with esa.EsaEngine(n, sigma) as eng:
        eng.set_sequence(enc)
        for k in ks:
            eng.set_prefixlength(k)
            eng.run(esa.WANT_SUF | esa.WANT_BCK)
            kk = eng.stats()["prefixlength"]
            assert k in (0, kk)
            got = eng.bcktab()
            want = ou.bcktab(enc, sigma, kk)
            for name, g, w in zip(("leftborder", "countspecialcodes", "distpfxidx"), got, want):
                assert np.array_equal(g, w), (name, kk)
            # buckets are what they claim: the suffixes between two borders
            # start with the bucket's k-mer
            if kk <= 5:
                suf = eng.table(esa.TAB_SUF)
                lb = got[0]
                for code in (0, len(lb) // 2, len(lb) - 2):
                    for i in range(int(lb[code]), min(int(lb[code + 1]), int(lb[code]) + 50)):
                        p = int(suf[i])
                        digits = [(code // sigma ** (kk - 1 - j)) % sigma for j in range(kk)]
                        seen = [int(x) for x in enc[p:p + kk]]
                        letters = next((j for j, x in enumerate(seen) if x >= 254), len(seen))
                        assert seen[:letters] == digits[:letters]
                        assert all(d == sigma - 1 for d in digits[letters:])


@pytest.mark.parametrize("wbits,n", [
    (15, 28_000),       # everything in one window, no partition pass
    (10, 40_000),       # one partition pass
    (4, 300_000),       # two passes
    (4, 1_500_000),     # two passes and two workgroups per (twice too large) window
    (15, 1_500_000),    # the shape a build of this size takes by itself
])
def test_rank_table_window_shapes(gpu, monkeypatch, wbits, n):
    """rank table through LDS windows (k_rank_window): GTAMD_RANK_WINDOW_BITS
    shrinks the window so that the partition shapes of a 3 Gbp build (two
    passes, split windows) are reached at test sizes; repeats make sure the
    doubling rounds really use the table"""
    monkeypatch.setenv("GTAMD_RANK_WINDOW_BITS", str(wbits))
    monkeypatch.setenv("GTAMD_NO_PAIRS", "1")   # (pairs would not need the table)
    enc = synth.generate(synth.MODEL_HUMANLIKE_DNA, 77 + wbits, n)
    if n < 200_000:     # too short for the model's repeats: a tandem repeat instead
        enc = np.resize(synth.generate(synth.MODEL_UNIFORM_DNA, 5, 700), n).astype(np.uint8)
    res = esa.suffixerator_tables(enc, 4)
    assert res.stats["refine_rounds"] > 0
    _assert_same_as_oracle(enc, 4, res)


@pytest.mark.parametrize("wbits,n,later", [(15, 3_000_000, False), (8, 600_000, False), (4, 600_000, False),
                                           (10, 900_000, True), (5, 900_000, True)])
@pytest.mark.parametrize("lds", ["1", "0"])
def test_rank_table_of_selected_windows(gpu, monkeypatch, wbits, n, later, lds):
    """only the windows of positions the rounds can touch are built (k_win_mark,
    k_win_filter with compact positions, k_rank_window per selected window): a random
    text with a few blocks in several copies -- the tied suffixes lie in a small part
    of the text; `later`: copies longer than the first rounds reach, so that windows
    are added between rounds (k_win_check); the last window of the text among them"""
    monkeypatch.setenv("GTAMD_RANK_WINDOW_BITS", str(wbits))
    monkeypatch.setenv("GTAMD_NO_PAIRS", "1")
    monkeypatch.setenv("GTAMD_WIN_FILTER_LDS", lds)     # (0: the bitmap from global memory)
    rng = np.random.default_rng(wbits * 1000 + n % 997)
    enc = rng.integers(0, 4, size=n).astype(np.uint8)
    blk = 40_000 if later else 3_000
    src = enc[1000:1000 + blk].copy()
    for at in (n // 3, n // 2 + 17, n - blk):        # (the last copy ends the text)
        enc[at:at + blk] = src
    enc[n // 5] = 254
    res = esa.suffixerator_tables(enc, 4)
    st = res.stats
    assert st["refine_rounds"] > 0
    assert 0 < st["rank_entries_built"] < (n + 1) // 2
    _assert_same_as_oracle(enc, 4, res)


@pytest.mark.parametrize("n,sigma", [(70_337, 4), (70_337 + 512, 4), (70_337, 20), (2948, 20)])
def test_tie_group_that_reaches_the_last_word_of_the_table(gpu, monkeypatch, n, sigma):
    """the largest suffixes tied -- a run of the largest letter, and nothing but the
    terminator behind them in the table -- with a table length that leaves the last
    64-entry chunk of the rank-table partition with fewer entries than its item number:
    the lane that held the chunk's tie-bitmap word sat out the branch in which the
    others read it (round 3, found by the fuzzer with small rank windows: the heads
    of those entries were the entries themselves, the rounds did not converge)"""
    if n == 2948:
        monkeypatch.setenv("GTAMD_RANK_WINDOW_BITS", "6")
    # (the whole table, as a text with ties everywhere gets it: its first partition pass
    # makes the heads on the fly)
    monkeypatch.setenv("GTAMD_RANK_ALL_WINDOWS", "1")
    rng = np.random.default_rng(n + sigma)
    enc = rng.integers(0, sigma - 1, size=n).astype(np.uint8)     # (the largest letter is kept out ...)
    run = 1219 if n < 5000 else 3000
    at = n // 4
    enc[at:at + run] = sigma - 1                                  # (... but for one long run)
    res = esa.suffixerator_tables(enc, sigma)
    assert res.stats["refine_rounds"] > 0
    _assert_same_as_oracle(enc, sigma, res)


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("n", [1, 15, 16, 17, 4095, 4096, 4097, 70_001, 1_000_003])
def test_keygen_with_and_without_fused_first_pass(gpu, monkeypatch, fused, n):
    """DNA keygen: fused with the sort's dcode pass (k_dc_hist_dna +
    k_keygen_pass0_dna, the default) and plain (k_keygen_dna + all six passes,
    GTAMD_FUSED_PASS0=0, the A/B switch) give the reference's tables; sizes
    around the 16-suffix thread groups and the 4096-suffix tiles, many specials"""
    monkeypatch.setenv("GTAMD_FUSED_PASS0", fused)
    enc = synth.generate(synth.MODEL_HUMANLIKE_DNA, 1000 + n, n)
    rng = np.random.default_rng(n)
    # more specials than the model has: runs and single ones, separators too
    for at in rng.integers(0, n, size=max(1, n // 200)):
        enc[at:at + int(rng.integers(1, 40))] = 254
    for at in rng.integers(1, max(2, n - 1), size=n // 5000):
        if 0 < at < n - 1 and enc[at - 1] != 255 and enc[at + 1] != 255:
            enc[at] = 255
    if enc[0] == 255:
        enc[0] = 0
    if enc[-1] == 255:
        enc[-1] = 0
    res = esa.suffixerator_tables(enc, 4)
    _assert_same_as_oracle(enc, 4, res)


@pytest.mark.parametrize("kind", ["all_wildcards", "alternating", "every_20th", "runs_of_19",
                                  "separators_every_21"])
def test_keygen_first_pass_special_heavy(gpu, kind):
    """tiles in which most or all suffixes have a special among their first 20
    symbols: the rare-class path of k_keygen_pass0_dna becomes the only path"""
    n = 3 * 4096 + 77
    rng = np.random.default_rng(5)
    enc = rng.integers(0, 4, size=n).astype(np.uint8)
    if kind == "all_wildcards":
        enc[:] = 254
    elif kind == "alternating":
        enc[::2] = 254
    elif kind == "every_20th":
        enc[19::20] = 254
    elif kind == "runs_of_19":
        enc[:] = 254
        enc[::20] = rng.integers(0, 4, size=enc[::20].size)
    else:
        enc[21::22] = 255
        enc[-1] = 0
    res = esa.suffixerator_tables(enc, 4)
    _assert_same_as_oracle(enc, 4, res)


# ---------------------------------------------------------------------------
# the pair path (tie groups of two settled by one text comparison) and the
# 64-bit position path (GTAMD_FORCE_WIDE=1: the kernels a part build of a
# sequence with n >= 2^32 runs, and the exchange machinery with one part)
# ---------------------------------------------------------------------------
def _pair_cases():
    rng = np.random.default_rng(19)
    a = rng.integers(0, 4, 6000, dtype=np.uint8)
    b = a.copy()
    b[::97] = (b[::97] + 1) & 3                     # mutated copy: pairs of all depths
    yield "copy_with_mutations", np.concatenate([a, [255], b]).astype(np.uint8)
    yield "exact_copy_then_end", np.concatenate([a, [1], a]).astype(np.uint8)
    yield "copy_before_wildcard", np.concatenate([a, [254], a, [254], [2] * 30]).astype(np.uint8)
    yield "three_copies", np.concatenate([a, [255], a, [255], a]).astype(np.uint8)   # no pairs
    c = np.concatenate([a[:3000], a[:3000][::-1]])
    yield "pairs_next_to_triples", np.concatenate([c, [255], c[:4000], [255], c[:1500]]).astype(np.uint8)
    yield "tandem", np.tile(a[:300], 9).astype(np.uint8)
    yield "copy_without_separator", np.concatenate([a, a[:5000], a[5100:]]).astype(np.uint8)


@pytest.mark.parametrize("name,enc", list(_pair_cases()), ids=[c[0] for c in _pair_cases()])
@pytest.mark.parametrize("pairs", ["on", "off"])
def test_pair_path(gpu, monkeypatch, name, enc, pairs):
    """with and without the pair path (GTAMD_NO_PAIRS=1: everything through
    prefix doubling) the tables are the oracle's"""
    if pairs == "off":
        monkeypatch.setenv("GTAMD_NO_PAIRS", "1")
    res = esa.suffixerator_tables(enc, 4)
    _assert_same_as_oracle(enc, 4, res)
    if pairs == "off":
        assert res.stats["pair_suffixes"] == 0
    elif name == "three_copies":
        assert res.stats["pair_suffixes"] < 100
    elif name != "tandem":
        assert res.stats["pair_suffixes"] > 0


def test_pair_path_large(gpu):
    """2 Mbp of the human-like model: most tied suffixes are pairs"""
    enc = synth.generate(synth.MODEL_HUMANLIKE_DNA, 12, 2_000_000)
    res = esa.suffixerator_tables(enc, 4)
    assert res.stats["pair_suffixes"] > res.stats["tied_suffixes"] // 2
    rc, where = ou.check_suffix_array(enc, res.suf)
    assert rc == 0, (rc, where)
    t = ou.tables_given_sa(enc, res.suf)
    assert np.array_equal(res.lcp, t["lcp"])
    assert np.array_equal(res.llv, t["llv"])
    assert np.array_equal(res.bwt, t["bwt"])


@pytest.mark.parametrize("name", ["Atinsert.fna", "Duplicate.fna", "RandomN.fna", "TTTN.fna",
                                  "sw100K1.fsa"])
def test_wide_positions_reference_fixtures(gpu, monkeypatch, name):
    monkeypatch.setenv("GTAMD_FORCE_WIDE", "1")
    e = GOLDEN[name]
    protein = e["alphabet"] == "protein"
    enc = ou.encode_fasta(ou.fixture_path(name), protein)
    res = esa.suffixerator_tables(enc, 20 if protein else 4)
    for tab in ("suf", "lcp", "llv", "bwt"):
        assert _md5(getattr(res, tab)) == e["tables"][tab]["md5"], tab


@pytest.mark.parametrize("model,sigma,n,seed", [
    (synth.MODEL_UNIFORM_DNA, 4, 300000, 21),
    (synth.MODEL_HUMANLIKE_DNA, 4, 600000, 3),
    (synth.MODEL_PROTEIN, 20, 200000, 5),
])
def test_wide_positions_synthetic(gpu, monkeypatch, model, sigma, n, seed):
    monkeypatch.setenv("GTAMD_FORCE_WIDE", "1")
    enc = synth.generate(model, seed, n)
    res = esa.suffixerator_tables(enc, sigma)
    _assert_same_as_oracle(enc, sigma, res)


@pytest.mark.parametrize("name,enc", list(_cases()) + list(_pair_cases()),
                         ids=[c[0] for c in _cases()] + [c[0] for c in _pair_cases()])
def test_wide_positions_edge_cases(gpu, monkeypatch, name, enc):
    monkeypatch.setenv("GTAMD_FORCE_WIDE", "1")
    res = esa.suffixerator_tables(enc, 4)
    _assert_same_as_oracle(enc, 4, res)


def test_repeat_heavy_model_device_equals_numpy(gpu):
    """the hard-case model (50 % copied blocks, satellite arrays of 10^5..10^6
    bases) on the device and in numpy, and its build at 20 Mbp checked with the
    linear-time checker (LCP values beyond 100 000: a dozen doubling rounds)"""
    import torch
    from genometools_amd import _lib
    n = 20_000_000
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    _lib.check(_lib.load().gtamd_synth_bytes(0, synth.MODEL_REPEAT_HEAVY, 9, n, buf.data_ptr()))
    enc = buf.cpu().numpy()
    for lo, hi in ((0, 300000), (2_200_000, 2_800_000), (n - 100000, n)):
        assert np.array_equal(enc[lo:hi], synth.generate(synth.MODEL_REPEAT_HEAVY, 9, n, lo, hi))
    assert len(synth.satellites(n)) == 4
    with esa.EsaEngine(n, 4) as eng:
        eng.set_sequence_device(buf.data_ptr(), n)
        eng.run()
        res = eng.result()
    assert res.stats["refine_rounds"] >= 12 and res.stats["maxbranchdepth"] > 100000
    rc, where = ou.check_suffix_array(enc, res.suf)
    assert rc == 0, (rc, where)
    t = ou.tables_given_sa(enc, res.suf)
    assert np.array_equal(res.lcp, t["lcp"])
    assert np.array_equal(res.llv, t["llv"])
    assert np.array_equal(res.bwt, t["bwt"])


def test_small_groups(gpu, monkeypatch):
    """tie groups of three and four (sorted by direct comparisons): a pair with
    a third suffix that shares 20 symbols only, three and four copies; with
    the path switched off (GTAMD_NO_SMALL_GROUPS=1) the same tables"""
    rng = np.random.default_rng(23)
    a = rng.integers(0, 4, 5000, dtype=np.uint8)
    third = np.concatenate([a[700:722], rng.integers(0, 4, 300, dtype=np.uint8)])   # 22 shared symbols
    enc = np.concatenate([a, [255], a, [254], third, [255], a[:2000], [255], a[100:1900], [255],
                          rng.integers(0, 4, 4000, dtype=np.uint8)]).astype(np.uint8)
    enc = np.concatenate([enc, enc[::-1][:3000]]).astype(np.uint8)
    ora = ou.esa(enc, 4)
    res = esa.suffixerator_tables(enc, 4)
    _assert_same_as_oracle(enc, 4, res, ora)
    monkeypatch.setenv("GTAMD_NO_SMALL_GROUPS", "1")
    _assert_same_as_oracle(enc, 4, esa.suffixerator_tables(enc, 4), ora)


def test_small_groups_too_deep_fall_back(gpu):
    """three copies of 40 000 bases: the comparisons of the small-group path
    give up beyond 2^15 symbols, those groups go through prefix doubling"""
    rng = np.random.default_rng(29)
    a = rng.integers(0, 4, 40000, dtype=np.uint8)
    enc = np.concatenate([a, [255], a, [255], a, rng.integers(0, 4, 30000, dtype=np.uint8)]).astype(np.uint8)
    res = esa.suffixerator_tables(enc, 4)
    assert res.stats["refine_rounds"] >= 10
    rc, where = ou.check_suffix_array(enc, res.suf)
    assert rc == 0, (rc, where)
    t = ou.tables_given_sa(enc, res.suf)
    assert np.array_equal(res.lcp, t["lcp"])
    assert np.array_equal(res.llv, t["llv"])
    assert np.array_equal(res.bwt, t["bwt"])


@pytest.mark.parametrize("mode,wgs", [("0", None), ("1", None), ("2", None), ("2", "64"), ("1", "100000")])
def test_table_entries_of_the_pairs_beside_the_rounds(gpu, monkeypatch, mode, wgs):
    """the table entries of the pairs and small groups are written on the second
    stream beside the doubling rounds (GTAMD_APPLY_EARLY=2, the default), from
    the pair path on (1) or behind the rounds on the main stream (0), by a
    capped or a full grid: the same tables, `.llv` and statistics every time --
    pairs with LCP values beyond the byte, small groups, groups that need rounds"""
    rng = np.random.default_rng(31)
    a = rng.integers(0, 4, 6000, dtype=np.uint8)
    b = rng.integers(0, 4, 900, dtype=np.uint8)
    enc = np.concatenate([a[:3000], [255], b, b, b, b, [254], a[:3000], [255], a[500:2500], [255],
                          a[3000:], a[5000:5600], np.zeros(300, dtype=np.uint8)]).astype(np.uint8)
    ora = ou.esa(enc, 4)
    monkeypatch.setenv("GTAMD_APPLY_EARLY", mode)
    if wgs is not None:
        monkeypatch.setenv("GTAMD_APPLY_WGS", wgs)
    res = esa.suffixerator_tables(enc, 4)
    assert res.stats["pair_suffixes"] > 0 and res.stats["refine_rounds"] > 0
    assert res.stats["largelcpvalues"] > 0
    _assert_same_as_oracle(enc, 4, res, ora)
