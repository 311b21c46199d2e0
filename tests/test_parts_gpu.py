"""The sharded build (lexicographic range parts; every part filters the suffixes
of its key range from the replicated text -- DNA -- or keys its own text tile and
sends the pairs to the range owners -- protein; rank table cut by text position,
queried and updated through alltoallv): R engine contexts on one device must
produce, slice by slice, exactly the tables of the single build."""
import numpy as np
import pytest

import oracle_util as ou
from genometools_amd import synth
from thread_comm import build_in_parts, build_sequences_in_parts

pytestmark = pytest.mark.gpu


def _compare(tabs, stats, ora):
    assert np.array_equal(tabs["suf"], ora["suf"]), "suf"
    assert np.array_equal(tabs["bwt"], ora["bwt"]), "bwt"
    assert np.array_equal(tabs["lcp"], ora["lcp"]), "lcp"
    assert np.array_equal(tabs["llv"], ora["llv"]), "llv"
    st = ora["stats"]
    assert stats["longest"] == st["longest"]
    assert stats["largelcpvalues"] == st["largelcpvalues"]
    assert stats["maxbranchdepth"] == st["maxbranchdepth"]
    assert stats["lcptabsum"] == int(st["lcptabsum"])


def _check(enc, sigma, parts):
    tabs, stats, per_part = build_in_parts(enc, sigma, parts)
    _compare(tabs, stats, ou.esa(enc, sigma))
    return per_part


@pytest.mark.parametrize("parts", [2, 3, 4, 8])
def test_humanlike_in_parts(gpu, parts):
    enc = synth.generate(synth.MODEL_HUMANLIKE_DNA, 5, 400000)
    per_part = _check(enc, 4, parts)
    # ties (repeats) are spread over the parts and need remote rank lookups
    assert sum(1 for s in per_part if s["tied_suffixes"] > 0) >= 2


@pytest.mark.parametrize("parts", [2, 5])
def test_uniform_and_protein_in_parts(gpu, parts):
    _check(synth.generate(synth.MODEL_UNIFORM_DNA, 6, 300000), 4, parts)
    _check(synth.generate(synth.MODEL_PROTEIN, 7, 200000), 20, parts)


def test_degenerate_inputs_in_parts(gpu):
    rng = np.random.default_rng(3)
    a = rng.integers(0, 4, 3000, dtype=np.uint8)
    cases = [np.zeros(4000, dtype=np.uint8),                       # one giant tie group
             np.full(500, 254, dtype=np.uint8),                    # only the special tail
             np.concatenate([a, [255], a, [254], a]).astype(np.uint8),
             np.tile(np.array([0, 1, 2], dtype=np.uint8), 2000),
             a[:5]]                                                # fewer suffixes than parts
    for enc in cases:
        _check(enc, 4, 4)
        _check(enc, 4, 8)


@pytest.mark.parametrize("parts", [2, 3, 8])
def test_wide_positions_in_parts(gpu, monkeypatch, parts):
    """the 64-bit position / rank kernels of a build with n >= 2^32, forced at
    oracle sizes (GTAMD_FORCE_WIDE=1)"""
    monkeypatch.setenv("GTAMD_FORCE_WIDE", "1")
    _check(synth.generate(synth.MODEL_HUMANLIKE_DNA, 5, 400000), 4, parts)
    _check(synth.generate(synth.MODEL_PROTEIN, 7, 150000), 20, parts)
    _check(np.zeros(3000, dtype=np.uint8), 4, parts)
    rng = np.random.default_rng(3)
    a = rng.integers(0, 4, 3000, dtype=np.uint8)
    _check(np.concatenate([a, [255], a, [254], a]).astype(np.uint8), 4, parts)


def test_part_contexts_are_reusable(gpu):
    """a short sequence, a longer one and the short one again through the same
    part contexts: every per-run buffer follows the run's own sizes"""
    encs = [synth.generate(synth.MODEL_HUMANLIKE_DNA, 11, 70000),
            synth.generate(synth.MODEL_HUMANLIKE_DNA, 12, 500000),
            synth.generate(synth.MODEL_UNIFORM_DNA, 13, 1000),
            synth.generate(synth.MODEL_HUMANLIKE_DNA, 11, 70000)]
    for enc, (tabs, stats, _) in zip(encs, build_sequences_in_parts(encs, 4, 3)):
        _compare(tabs, stats, ou.esa(enc, 4))


def test_part_work_is_a_share_of_the_whole(gpu):
    """a part sorts its own slice only: the device memory it holds is about 1/R of
    the single build's (at a size where the fixed tables of the MSD sort -- 65 536
    histogram rows, 67 MB -- do not weigh)"""
    enc = synth.generate(synth.MODEL_UNIFORM_DNA, 4, 48_000_000)
    from genometools_amd import esa
    whole = esa.suffixerator_tables(enc, 4)
    tabs, stats, per_part = build_in_parts(enc, 4, 4)
    assert np.array_equal(tabs["suf"], whole.suf)
    for st in per_part:
        assert st["device_bytes"] < 0.4 * whole.stats["device_bytes"]


def test_many_parts(gpu):
    """more parts than the ballot loops unroll for; parts with empty slices"""
    _check(synth.generate(synth.MODEL_HUMANLIKE_DNA, 9, 120000), 4, 16)
    _check(np.tile(np.array([0, 1, 2, 3, 3], dtype=np.uint8), 400), 4, 6)


@pytest.mark.parametrize("parts", [1, 3])
def test_deep_groups_reach_new_rank_windows(gpu, monkeypatch, parts):
    """five copies of 30 000 bases: groups of five stay tied for 11 rounds and
    the offsets of the late rounds (h > 10 240) reach windows of the rank table
    that were not built / sent at first -- the lazy extension, in a single
    build (small windows forced) and in a part build"""
    rng = np.random.default_rng(31)
    a = rng.integers(0, 4, 30000, dtype=np.uint8)
    enc = np.concatenate([a, [255], a, [254], a, [255], a, [255], a,
                          rng.integers(0, 4, 200000, dtype=np.uint8)]).astype(np.uint8)
    if parts == 1:
        from genometools_amd import esa
        monkeypatch.setenv("GTAMD_RANK_WINDOW_BITS", "8")
        res = esa.suffixerator_tables(enc, 4)
        tabs = {"suf": res.suf, "lcp": res.lcp, "llv": res.llv, "bwt": res.bwt}
        rounds = res.stats["refine_rounds"]
    else:
        tabs, stats, _ = build_in_parts(enc, 4, parts)
        rounds = stats["refine_rounds"]
    assert rounds >= 10
    rc, where = ou.check_suffix_array(enc, tabs["suf"])
    assert rc == 0, (rc, where)
    t = ou.tables_given_sa(enc, tabs["suf"])
    assert np.array_equal(tabs["lcp"], t["lcp"])
    assert np.array_equal(tabs["llv"], t["llv"])
    assert np.array_equal(tabs["bwt"], t["bwt"])


@pytest.mark.parametrize("parts", [2, 3])
def test_pairs_in_some_parts_only(gpu, parts):
    """Two copies of a block over {A, C} in a text over {G, T}: deep pairs (too
    deep for the direct comparison of a few ties), all in the lowest range, so
    the other parts have no pair list at all -- and every step the parts agree
    on must still be taken by all of them (found by the fuzz campaign: a part
    without pairs skipped one and the build stopped)."""
    rng = np.random.default_rng(12)        # (this seed: no 20 symbols twice outside the block)
    a = rng.integers(2, 4, 1500, dtype=np.uint8)
    blk = rng.integers(0, 2, 300, dtype=np.uint8)
    enc = np.concatenate([a[:500], blk, [2], a[500:1000], blk, [3], a[1000:]]).astype(np.uint8)
    per_part = _check(enc, 4, parts)
    assert sum(1 for s in per_part if s["pair_suffixes"] > 0) == 1
    assert sum(1 for s in per_part if s["tied_suffixes"] == 0) >= 1


@pytest.mark.timeout(120)
def test_a_failing_part_does_not_leave_the_others_waiting(gpu):
    """the library's thread transport (csrc/esa_comm.hip): a part whose run fails --
    here before its first collective: it never got a sequence -- tells the transport
    (gtamd_esa_set_comm_abort, registered by gtamd_comm_attach), and the other part
    returns -1 from its next collective instead of blocking in it"""
    import threading
    from genometools_amd import _lib, esa
    lib = _lib.load()
    enc = synth.generate(synth.MODEL_HUMANLIKE_DNA, 9, 300000)
    comm = lib.gtamd_comm_threads_create(2)
    assert comm
    ctxs = [lib.gtamd_esa_create(0, enc.size, 4) for _ in range(2)]
    try:
        assert all(ctxs)
        for r, c in enumerate(ctxs):
            assert lib.gtamd_comm_attach(comm, r, c, 0) == 0
        assert lib.gtamd_esa_set_sequence_bytes(ctxs[0], enc.ctypes.data, enc.size, 0) == 0
        rc = [None, None]

        def run(r):
            rc[r] = lib.gtamd_esa_run(ctxs[r], esa.WANT_SUF | esa.WANT_LCP)

        th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(60)
        assert not any(t.is_alive() for t in th), "a part is still waiting"
        assert rc == [-1, -1]
    finally:
        for c in ctxs:
            if c:
                lib.gtamd_esa_destroy(c)
        lib.gtamd_comm_destroy(comm)
