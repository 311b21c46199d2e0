"""The device packed-index builder (include/gtamd_pck.h) against INDEX.bdx files
the reference wrote (tests/golden/golden_pck.json) and against the CPU
restatement (oracle/pck_oracle.c) on synthetic sequences; through the C ABI."""
import hashlib

import numpy as np
import pytest

import oracle_util as ou
from genometools_amd import esa, pck, synth

pytestmark = pytest.mark.gpu

GOLDEN = ou.golden_pck()


def _names():
    return sorted({k.split("|")[0] for k in GOLDEN})


@pytest.fixture(scope="module")
def builder():
    with pck.PackedIndex() as p:
        yield p


@pytest.mark.parametrize("name", _names())
def test_device_bdx_equals_reference_files(name, builder):
    protein = name.endswith((".fsa", ".faa"))
    sigma = 20 if protein else 4
    enc = ou.encode_fasta(ou.fixture_path(name), protein)
    keys = sorted(k for k in GOLDEN if k.split("|")[0] == name)
    for direction in sorted({ou.parse_pck_key(k)[1].get("direction", "fwd") for k in keys}):
      with esa.EsaEngine(enc.size, sigma) as eng:
        # the project as `gt suffixerator -dir DIRECTION` builds it
        eng.set_readmode({"fwd": 0, "rev": 1, "cpl": 2, "rcl": 3}[direction])
        eng.set_sequence(enc)
        eng.run(esa.WANT_SUF | esa.WANT_BWT)
        for key in keys:
            _, kw = ou.parse_pck_key(key)
            if kw.pop("direction", "fwd") != direction:
                continue
            builder.build_from_esa(eng, **kw)
            raw = builder.image().tobytes()
            e = GOLDEN[key]
            assert pck.default_toggles(kw["bsize"], kw["blbuck"], kw["locfreq"], kw["locbitmap"],
                                       kw.get("sprank", False)) == e["featureToggles"], key
            assert len(raw) == e["size"], key
            assert hashlib.md5(raw).hexdigest() == e["md5"], key


def _against_oracle(enc, sigma, builder, **kw):
    with esa.EsaEngine(enc.size, sigma) as eng:
        eng.set_sequence(enc)
        eng.run(esa.WANT_SUF | esa.WANT_BWT)
        suf, bwt = eng.table(esa.TAB_SUF), eng.table(esa.TAB_BWT)
        builder.build_from_esa(eng, **kw)
        got = builder.image().tobytes()
        want = ou.pck_bdx(enc, sigma, suf, bwt, **kw)
        if got != want:
            n = min(len(got), len(want))
            diff = [i for i in range(n) if got[i] != want[i]]
            inf = builder.info()
            raise AssertionError("%d / %d bytes, %d differ, first at %s; layout %s" % (
                len(got), len(want), len(diff), diff[:8], inf))


@pytest.mark.parametrize("model,sigma,n", [
    (synth.MODEL_HUMANLIKE_DNA, 4, 300000), (synth.MODEL_UNIFORM_DNA, 4, 65536 * 3),
    (synth.MODEL_HUMANLIKE_DNA, 4, 64 * 1000 - 1), (synth.MODEL_HUMANLIKE_DNA, 4, 64 * 1000),
    (synth.MODEL_HUMANLIKE_DNA, 4, 64 * 1000 + 1), (synth.MODEL_PROTEIN, 20, 100000)])
def test_device_bdx_equals_oracle_on_synthetic(model, sigma, n, builder):
    enc = synth.generate(model, 7, n)
    sets = [dict(), dict(locbitmap=True), dict(locfreq=0), dict(mkindex=True),
            dict(mkindex=True, locbitmap=True, locfreq=5), dict(sprank=True),
            dict(sprank=True, mkindex=True, bsize=10, locfreq=32),
            dict(sprank=True, locbitmap=True, locfreq=3)] if sigma == 4 else \
        [dict(bsize=1), dict(bsize=2, blbuck=5, locbitmap=True), dict(bsize=3, blbuck=3, locfreq=0),
         dict(bsize=3, mkindex=True), dict(bsize=2, sprank=True)]
    for kw in sets:
        _against_oracle(enc, sigma, builder, **kw)


def test_geometries_and_bucket_borders(builder):
    """block sizes and bucket lengths that do not divide the tile, sequences
    that end on / just before / just behind a bucket border, buckets longer
    than a wave, a locate interval that is no power of two"""
    rng = np.random.default_rng(5)
    for n in (1, 2, 63, 64, 65, 4095, 4096, 20000):
        enc = rng.integers(0, 4, n).astype(np.uint8)
        if n > 100:
            enc[rng.integers(0, n, n // 50)] = 254
            enc[rng.integers(0, n, 5)] = 255
            enc[n // 3:n // 3 + 70] = 254
        for kw in (dict(), dict(bsize=3, blbuck=5, locfreq=7, locbitmap=True),
                   dict(bsize=5, blbuck=2, locfreq=1, locbitmap=False),
                   dict(bsize=1, blbuck=1, locfreq=3), dict(bsize=12, blbuck=100, locfreq=32),
                   dict(mkindex=True), dict(bsize=4, blbuck=3, locfreq=6, mkindex=True),
                   dict(sprank=True), dict(sprank=True, bsize=3, blbuck=5, locfreq=7, locbitmap=False),
                   dict(bsize=16, blbuck=2, locfreq=0), dict(bsize=2, blbuck=4096, locfreq=16,
                                                            locbitmap=True)):
            _against_oracle(enc, 4, builder, **kw)


def test_raw_pointer_entry_point_and_errors(builder):
    enc = synth.generate(synth.MODEL_HUMANLIKE_DNA, 3, 50000)
    with esa.EsaEngine(enc.size, 4) as eng:
        eng.set_sequence(enc)
        eng.run(esa.WANT_SUF | esa.WANT_BWT)
        st = eng.stats()
        builder.build(eng.device_pointer(esa.TAB_BWT), eng.device_pointer(esa.TAB_SUF),
                      enc.size + 1, 4, st["longest"])
        a = builder.image().tobytes()
        builder.build_from_esa(eng)
        assert a == builder.image().tobytes()
        inf = builder.info()
        assert inf["file_bytes"] == len(a) and inf["num_buckets"] == (enc.size + 2 + 63) // 64
        # partial reads of the image
        assert builder.image(100, 50).tobytes() == a[100:150]
        with pytest.raises(esa.EsaError):
            builder.image(len(a) - 1, 2)
        with pytest.raises(esa.EsaError):
            builder.build_from_esa(eng, bsize=17)
        with pytest.raises(esa.EsaError):
            builder.build_from_esa(eng, bsize=8, blbuck=4096)
        eng.run(esa.WANT_SUF)
        with pytest.raises(esa.EsaError):
            builder.build_from_esa(eng)


CTX = ou.golden_ctxmap()


def test_device_context_maps_equal_reference_files(builder):
    """INDEX.<I>cxm (-ctxilog I, `gt packedindex mkctxmap`) from the engine's suffix array"""
    done = 0
    for name in sorted({k.split("|")[0] for k in CTX}):
        protein = name.endswith((".fsa", ".faa"))
        enc = ou.encode_fasta(ou.fixture_path(name), protein)
        keys = [k for k in sorted(CTX) if k.split("|")[0] == name]
        for direction in sorted({dict(p.split("=") for p in k.split("|")[1:]).get("dir", "fwd")
                                 for k in keys}):
            with esa.EsaEngine(enc.size, 20 if protein else 4) as eng:
                eng.set_readmode({"fwd": 0, "rev": 1}[direction])
                eng.set_sequence(enc)
                eng.run(esa.WANT_SUF)
                for key in keys:
                    kw = dict(p.split("=") for p in key.split("|")[1:])
                    if kw.get("dir", "fwd") != direction:
                        continue
                    used, raw = builder.context_map_from_esa(eng, int(kw["ctxilog"]))
                    e = CTX[key]
                    assert used == e["used"], key
                    assert raw.size == e["size"], key
                    assert hashlib.md5(raw.tobytes()).hexdigest() == e["md5"], key
                    done += 1
    assert done == len(CTX)


def test_context_map_on_synthetic_and_invalid_interval(builder):
    for n in (100, 4096, 300000):
        enc = synth.generate(synth.MODEL_HUMANLIKE_DNA, 5, n)
        with esa.EsaEngine(enc.size, 4) as eng:
            eng.set_sequence(enc)
            eng.run(esa.WANT_SUF)
            suf = eng.table(esa.TAB_SUF)
            for ilog in (-1, 0, 2, 6):
                used, raw = builder.context_map_from_esa(eng, ilog)
                want_used, want = ou.pck_ctxmap(suf, ilog)
                assert used == want_used and raw.tobytes() == want, (n, ilog)
            with pytest.raises(esa.EsaError):
                builder.context_map_from_esa(eng, 40)     # interval longer than the sequence
