"""The first sort of a DNA whole-table build, most significant digit first
(genometools_amd/csrc/esa_msd.h): keygen fused with level A, two ragged-tile
levels, the LDS sort that also emits the tables, the one-workgroup path for
oversize runs and the device-wide path for giant ones.  By default the engine
takes it from 2^25 entries; GTAMD_MSD=1 takes it at any size, so that all of it
is checked bit for bit against the oracle and the reference's fixtures, and at
sizes the oracle cannot reach against the LSD sort (GTAMD_MSD=0)."""
import hashlib

import numpy as np
import pytest

import oracle_util as ou
from genometools_amd import esa, synth
from test_esa_gpu import _assert_same_as_oracle, _cases, _pair_cases

pytestmark = pytest.mark.gpu
GOLDEN = ou.golden()
DNA_FIXTURES = sorted(k for k in GOLDEN if GOLDEN[k]["alphabet"] != "protein")


def _md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture
def msd(monkeypatch):
    monkeypatch.setenv("GTAMD_MSD", "1")
    return monkeypatch


@pytest.mark.parametrize("name", DNA_FIXTURES)
def test_reference_fixtures(gpu, msd, name):
    e = GOLDEN[name]
    enc = ou.encode_fasta(ou.fixture_path(name), False)
    res = esa.suffixerator_tables(enc, 4)
    assert _md5(res.suf) == e["tables"]["suf"]["md5"]
    assert _md5(res.lcp) == e["tables"]["lcp"]["md5"]
    assert _md5(res.llv) == e["tables"]["llv"]["md5"]
    assert _md5(res.bwt) == e["tables"]["bwt"]["md5"]
    ss = ou.seqstats(enc, 4)
    assert esa.prj_text(ss, res.stats) == e["prj"]


@pytest.mark.parametrize("cbits", ["0", "3", "8"])
@pytest.mark.parametrize("n", [63, 64, 65, 255, 4095, 4096, 4097, 8191, 20000, 70001])
def test_uniform_dna_small(gpu, msd, cbits, n):
    msd.setenv("GTAMD_MSD_CBITS", cbits)
    enc = synth.generate(synth.MODEL_UNIFORM_DNA, 42, n)
    res = esa.suffixerator_tables(enc, 4)
    _assert_same_as_oracle(enc, 4, res)


@pytest.mark.parametrize("radix", ["0", "1"])
@pytest.mark.parametrize("cbits", ["0", "5", "8"])
@pytest.mark.parametrize("model,n,seed", [
    (synth.MODEL_UNIFORM_DNA, 1 << 20, 1),
    (synth.MODEL_HUMANLIKE_DNA, 70000, 2),
    (synth.MODEL_HUMANLIKE_DNA, 600000, 3),
    (synth.MODEL_REPEAT_HEAVY, 300000, 6),
])
def test_synthetic_models(gpu, msd, radix, cbits, model, n, seed):
    """radix=1: the LSD passes inside every run (what a run with a crowded bin
    of the counting pass falls back to)"""
    msd.setenv("GTAMD_MSD_CBITS", cbits)
    msd.setenv("GTAMD_MSD_RADIX", radix)
    enc = synth.generate(model, seed, n)
    res = esa.suffixerator_tables(enc, 4)
    _assert_same_as_oracle(enc, 4, res)


@pytest.mark.parametrize("name,enc", list(_cases()) + list(_pair_cases()),
                         ids=[c[0] for c in list(_cases()) + list(_pair_cases())])
@pytest.mark.parametrize("cbits", ["0", "8"])
def test_edge_cases(gpu, msd, cbits, name, enc):
    """one letter, periods, specials everywhere: runs above the LDS tile (one
    workgroup each)"""
    msd.setenv("GTAMD_MSD_CBITS", cbits)
    res = esa.suffixerator_tables(enc, 4)
    _assert_same_as_oracle(enc, 4, res)


def _skewed():
    rng = np.random.default_rng(23)
    r = rng.integers(0, 4, 30000, dtype=np.uint8)
    yield "poly_A_inside_random", np.concatenate([r, np.zeros(20000, np.uint8), r[:5000]])
    yield "poly_T_at_the_end", np.concatenate([r, np.full(9000, 3, np.uint8)])
    yield "wildcard_run", np.concatenate([r, np.full(15000, 254, np.uint8), r[:100]])
    yield "two_letters", rng.integers(0, 2, 60000, dtype=np.uint8)
    yield "one_12mer_over_and_over", np.concatenate(
        [np.concatenate([r[:12], rng.integers(0, 4, 9, dtype=np.uint8)]) for _ in range(3000)])
    yield "separators_everywhere", np.where(rng.random(50000) < 0.1, 255,
                                            rng.integers(0, 4, 50000)).astype(np.uint8)


@pytest.mark.parametrize("name,enc", list(_skewed()), ids=[c[0] for c in _skewed()])
@pytest.mark.parametrize("big_max", ["4096", "524288"])
@pytest.mark.parametrize("cbits", ["0", "8"])
def test_skewed_ranges(gpu, msd, cbits, big_max, name, enc):
    """ranges far above the LDS tile; with GTAMD_MSD_BIG_MAX=4096 every oversize
    run takes the device-wide path of the giant runs"""
    msd.setenv("GTAMD_MSD_CBITS", cbits)
    msd.setenv("GTAMD_MSD_BIG_MAX", big_max)
    res = esa.suffixerator_tables(enc, 4)
    _assert_same_as_oracle(enc, 4, res)


@pytest.mark.parametrize("kind", ["all_wildcards", "alternating", "every_20th", "runs_of_19",
                                  "separators_every_21"])
def test_special_heavy(gpu, msd, kind):
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


def test_want_subsets_and_reuse(gpu, msd):
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


@pytest.mark.parametrize("model,n,seed", [
    (synth.MODEL_HUMANLIKE_DNA, 40_000_000, 11),     # 2 bits at level C
    (synth.MODEL_REPEAT_HEAVY, 20_000_003, 12),      # 1 bit, satellite arrays: big runs
    (synth.MODEL_UNIFORM_DNA, 150_000_000, 13),      # 4 bits
])
def test_same_tables_as_the_lsd_sort(gpu, monkeypatch, model, n, seed):
    """beyond the oracle's reach: the two sorts must agree on every table and
    statistic (the LSD sort is pinned at these sizes by test_fullsize_gpu and by
    the property tests of test_esa_gpu)"""
    enc = synth.generate(model, seed, n)
    out = {}
    with esa.EsaEngine(n, 4) as eng:
        eng.set_sequence(enc)
        for mode in ("0", "1"):
            monkeypatch.setenv("GTAMD_MSD", mode)
            eng.run(esa.WANT_SUF | esa.WANT_LCP | esa.WANT_BWT)
            r = eng.result()
            out[mode] = (_md5(r.suf), _md5(r.lcp), _md5(r.bwt), _md5(r.llv), dict(r.stats))
    assert out["0"][:4] == out["1"][:4]
    for k in ("longest", "largelcpvalues", "maxbranchdepth", "lcptabsum", "prefixlength",
              "numberofallsortedsuffixes"):
        assert out["0"][4][k] == out["1"][4][k], k


@pytest.mark.parametrize("pack", ["0", "1"])
@pytest.mark.parametrize("cbits", ["0", "4", "8"])
def test_level_d_tiles_packed_and_by_stride(gpu, msd, pack, cbits):
    """the runs of level D as whole ranges packed into tiles (default) and cut by
    the stride rule (GTAMD_MSD_PACK=0, kept for comparison); a skewed composition
    makes ranges of very different sizes, among them ranges above the LDS tile"""
    msd.setenv("GTAMD_MSD_PACK", pack)
    msd.setenv("GTAMD_MSD_CBITS", cbits)
    rng = np.random.default_rng(17)
    for n, p in ((150000, [0.45, 0.05, 0.05, 0.45]), (400000, [0.7, 0.1, 0.1, 0.1])):
        enc = rng.choice(4, size=n, p=p).astype(np.uint8)
        enc[rng.integers(0, n, 40)] = 254
        res = esa.suffixerator_tables(enc, 4)
        _assert_same_as_oracle(enc, 4, res)


def test_depth_of_level_c_is_chosen_from_the_ranges(gpu, msd):
    """without GTAMD_MSD_CBITS the depth comes from the sizes of the ranges level B
    leaves (k_msd_skew); whatever it picks, the tables are the oracle's"""
    rng = np.random.default_rng(23)
    for p in ([0.25, 0.25, 0.25, 0.25], [0.4, 0.1, 0.1, 0.4]):
        enc = rng.choice(4, size=300000, p=p).astype(np.uint8)
        res = esa.suffixerator_tables(enc, 4)
        _assert_same_as_oracle(enc, 4, res)


# ---------------------------------------------------------------------------
# the 5-bit alphabets through the same levels (esa_msd.h, FMT 1: a 40-bit code of
# nine symbols -- four pairs as 9-bit numbers, the ninth by its class)
# ---------------------------------------------------------------------------
PROTEIN_FIXTURES = sorted(k for k in GOLDEN if GOLDEN[k]["alphabet"] == "protein")


@pytest.mark.parametrize("name", PROTEIN_FIXTURES)
def test_protein_reference_fixtures(gpu, msd, name):
    e = GOLDEN[name]
    enc = ou.encode_fasta(ou.fixture_path(name), True)
    res = esa.suffixerator_tables(enc, 20)
    assert _md5(res.suf) == e["tables"]["suf"]["md5"]
    assert _md5(res.lcp) == e["tables"]["lcp"]["md5"]
    assert _md5(res.llv) == e["tables"]["llv"]["md5"]
    assert _md5(res.bwt) == e["tables"]["bwt"]["md5"]
    ss = ou.seqstats(enc, 20)
    assert esa.prj_text(ss, res.stats) == e["prj"]


def _protein_cases():
    rng = np.random.default_rng(77)
    yield "model_70k", synth.generate(synth.MODEL_PROTEIN, 3, 70001)
    yield "model_1m", synth.generate(synth.MODEL_PROTEIN, 4, 1 << 20)
    r = rng.integers(0, 20, 30000, dtype=np.uint8)
    yield "copies", np.concatenate([r, [255], r[:20000], [254], r[5000:]]).astype(np.uint8)
    yield "one_letter", np.full(20000, 7, dtype=np.uint8)
    yield "two_letters", rng.integers(18, 20, 50000, dtype=np.uint8)          # the largest letters
    yield "period_3", np.tile(np.array([0, 19, 5], dtype=np.uint8), 9000)
    yield "all_wildcards", np.full(5000, 254, dtype=np.uint8)
    yield "specials_everywhere", np.where(rng.random(60000) < 0.12, 254 + (rng.random(60000) < 0.5),
                                          rng.integers(0, 20, 60000)).astype(np.uint8)
    x = rng.integers(0, 20, 40000).astype(np.uint8)
    x[8::9] = 255                                   # a special behind every eight letters
    x[-1] = 3
    yield "separator_every_9th", x
    y = rng.integers(0, 20, 40000).astype(np.uint8)
    y[9::10] = 254                                  # ... behind every nine: the class of the ninth
    yield "wildcard_every_10th", y
    yield "tiny", np.array([3, 19, 254, 0, 0], dtype=np.uint8)


@pytest.mark.parametrize("name,enc", list(_protein_cases()), ids=[c[0] for c in _protein_cases()])
@pytest.mark.parametrize("cbits", ["0", "8"])
@pytest.mark.parametrize("big_max", ["4096", "524288"])
def test_protein_cases(gpu, msd, big_max, cbits, name, enc):
    msd.setenv("GTAMD_MSD_CBITS", cbits)
    msd.setenv("GTAMD_MSD_BIG_MAX", big_max)
    res = esa.suffixerator_tables(enc, 20)
    _assert_same_as_oracle(enc, 20, res)
    if enc.size >= 64:
        assert res.timing["dominant_kernel"] == 1      # (it WAS the MSD sort)


@pytest.mark.parametrize("radix", ["0", "1"])
def test_protein_radix_fallback_and_prefixlength(gpu, msd, radix):
    """the statistics of .prj mask the LCP sum with the prefix length: the code
    must tell "at least k letters" for every k it is used with (k <= 9)"""
    import ctypes
    msd.setenv("GTAMD_MSD_RADIX", radix)
    rng = np.random.default_rng(5)
    enc = synth.generate(synth.MODEL_PROTEIN, 9, 300000)
    enc[rng.integers(0, enc.size, 3000)] = 254        # many short stretches of letters
    ora = ou.esa(enc, 20)
    L = ou.lib()
    for pl in (0, 1, 5, 8, 9):
        with esa.EsaEngine(enc.size, 20) as eng:
            eng.set_prefixlength(pl)
            eng.set_sequence(enc)
            eng.run()
            res = eng.result()
        assert res.timing["dominant_kernel"] == 1
        assert np.array_equal(res.suf, ora["suf"]) and np.array_equal(res.lcp, ora["lcp"])
        assert np.array_equal(res.bwt, ora["bwt"])
        st = ou.EsaStats()
        L.ora_esastats_compute(ou._p(enc), enc.size, ou._p(ora["suf"]), ou._p(ora["lcpfull"]),
                               pl or res.stats["prefixlength"], ctypes.byref(st))
        assert res.stats["lcptabsum"] == int(st.lcptabsum), pl
        assert res.stats["maxbranchdepth"] == st.maxbranchdepth


def test_protein_same_tables_as_the_lsd_sort(gpu, monkeypatch):
    enc = synth.generate(synth.MODEL_PROTEIN, 21, 30_000_000)
    out = {}
    with esa.EsaEngine(enc.size, 20) as eng:
        eng.set_sequence(enc)
        for mode in ("0", "1"):
            monkeypatch.setenv("GTAMD_MSD", mode)
            eng.run(esa.WANT_SUF | esa.WANT_LCP | esa.WANT_BWT)
            r = eng.result()
            out[mode] = (_md5(r.suf), _md5(r.lcp), _md5(r.bwt), _md5(r.llv), dict(r.stats))
    assert out["0"][:4] == out["1"][:4]
    for k in ("longest", "largelcpvalues", "maxbranchdepth", "lcptabsum", "prefixlength"):
        assert out["0"][4][k] == out["1"][4][k], k
