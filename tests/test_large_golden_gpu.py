"""The engine's DEFAULT code path at the sizes where it switches (the MSD first sort
from 2^25 entries, the pair path, the windowed rank table, the doubling rounds)
against tables written by the REFERENCE itself:

* tests/golden/golden_large.json -- md5 + size of INDEX.suf/.lcp/.llv/.bwt and the
  INDEX.prj text of oracle/_ref/gt_ref_sfx (the reference's Sfxiterator, compiled
  from /root/reference by oracle/Makefile.ref) on seeded synthetic sequences
  (tests/golden/make_golden_large.py; the device generator makes the same bytes);
  every case as ONE build and in THREE parts (range filter + MSD sort per part,
  ranks exchanged through the thread transport);
* the 256 Mbp input of BASELINE.md section 2 (CPython random.seed(42), regenerated
  here draw by draw) against the md5 sums BASELINE.md publishes for the tables the
  survey session made with the reference's own build.

No environment switch is set: whatever the engine takes by default at these sizes
is what is compared, entry for entry (md5 over the whole table).
"""
import hashlib
import json
import os
import random

import numpy as np
import pytest
import torch

import oracle_util as ou
from genometools_amd import _lib, esa, synth
from thread_comm import build_in_parts

pytestmark = pytest.mark.gpu

with open(os.path.join(ou.GOLDEN_DIR, "golden_large.json")) as _f:
    LARGE = json.load(_f)


def _md5(a):
    return hashlib.md5(np.ascontiguousarray(a).data).hexdigest()


def _device_sequence(model, seed, n):
    lib = _lib.load()
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    _lib.check(lib.gtamd_synth_bytes(0, model, seed, n, buf.data_ptr()))
    torch.cuda.synchronize()
    return buf


def _prj(text):
    return dict(line.split("=") for line in text.splitlines())


def _check_tables(e, suf, lcp, llv, bwt):
    for name, tab in (("suf", suf), ("lcp", lcp), ("llv", llv), ("bwt", bwt)):
        assert tab.nbytes == e["tables"][name]["bytes"], name
        assert _md5(tab) == e["tables"][name]["md5"], name


def _check_stats(e, stats):
    prj = _prj(e["prj"])
    n1 = int(prj["numberofallsortedsuffixes"])
    assert stats["numberofallsortedsuffixes"] == n1
    assert stats["longest"] == int(prj["longest"])
    assert stats["prefixlength"] == int(prj["prefixlength"])
    assert stats["largelcpvalues"] == int(prj["largelcpvalues"])
    assert stats["maxbranchdepth"] == int(prj["maxbranchdepth"])
    assert "%.2f" % (stats["lcptabsum"] / n1) == prj["averagelcp"]


@pytest.mark.parametrize("name", sorted(LARGE))
def test_single_build_equals_reference_tables(gpu, name):
    e = LARGE[name]
    n, sigma = e["n"], 20 if e["alphabet"] == "protein" else 4
    buf = _device_sequence(getattr(synth, e["model"]), e["seed"], n)
    with esa.EsaEngine(n, sigma) as eng:
        eng.set_sequence_device(buf.data_ptr(), n)
        del buf
        eng.run()
        stats = eng.stats()
        _check_tables(e, eng.table(esa.TAB_SUF), eng.table(esa.TAB_LCP), eng.table(esa.TAB_LLV),
                      eng.table(esa.TAB_BWT))
    _check_stats(e, stats)
    if sigma == 4 and n + 1 >= 1 << 25:
        # (what was compared IS the MSD path: it reports its runs)
        assert stats["msd_big_entries"] + stats["msd_crowded_entries"] > 0 or "uniform" in name


@pytest.mark.parametrize("name", sorted(k for k in LARGE if LARGE[k]["n"] <= 64 * 1000 * 1000))
def test_three_parts_equal_reference_tables(gpu, name):
    e = LARGE[name]
    n, sigma = e["n"], 20 if e["alphabet"] == "protein" else 4
    enc = _device_sequence(getattr(synth, e["model"]), e["seed"], n).cpu().numpy()
    tabs, stats, per_part = build_in_parts(enc, sigma, 3)
    _check_tables(e, tabs["suf"], tabs["lcp"], tabs["llv"], tabs["bwt"])
    prj = _prj(e["prj"])
    assert stats["longest"] == int(prj["longest"])
    assert stats["largelcpvalues"] == int(prj["largelcpvalues"])
    assert stats["maxbranchdepth"] == int(prj["maxbranchdepth"])
    assert "%.2f" % (stats["lcptabsum"] / (n + 1)) == prj["averagelcp"]
    assert all(s["numberofallsortedsuffixes"] == n + 1 for s in per_part)


def _baseline_md_sequence(n, seed=42, chunk=1000000):
    """`random.seed(42); random.choices('ACGT', k=1000000)` per
    megabase.  random.choices draws floor(random() * 4) per symbol; numpy's legacy
    generator is the same MT19937 with the same 53-bit doubles, so the draws are
    replayed from the state CPython's seeding leaves (checked against
    random.choices itself on the first megabase below)."""
    random.seed(seed)
    out = np.empty(n, dtype=np.uint8)
    for i in range(0, n, chunk):
        st = random.getstate()
        rs = np.random.RandomState()
        rs.set_state(("MT19937", np.array(st[1][:-1], dtype=np.uint32), st[1][-1]))
        k = min(chunk, n - i)
        out[i:i + k] = np.floor(rs.random_sample(k) * 4)
        ns = rs.get_state()
        random.setstate((st[0], tuple(int(v) for v in ns[1]) + (int(ns[2]),), st[2]))
    return out


BASELINE_MD = {"suf": "963e70bd37858e9d1f4ac254ae58cfd9", "lcp": "e239f1f031c9b06f3ecf9655441b83dc",
               "bwt": "f00eb2231c0ef1a058eef5fc20183a58"}


def test_baseline_md_256mbp_input_equals_published_md5(gpu):
    """the numbers BASELINE.md quotes for this input: prefixlength=11,
    averagelcp=13.16, maxbranchdepth=27, largelcpvalues=0"""
    n = 256 * 1000 * 1000
    enc = _baseline_md_sequence(n)
    random.seed(42)
    first = np.frombuffer("".join(random.choices("ACGT", k=1000000)).encode(), dtype=np.uint8)
    code = np.zeros(256, dtype=np.uint8)
    code[ord("C")], code[ord("G")], code[ord("T")] = 1, 2, 3
    assert np.array_equal(code[first], enc[:1000000])
    with esa.EsaEngine(n, 4) as eng:
        eng.set_sequence(enc)
        del enc
        eng.run()
        stats = eng.stats()
        assert _md5(eng.table(esa.TAB_SUF)) == BASELINE_MD["suf"]
        assert _md5(eng.table(esa.TAB_LCP)) == BASELINE_MD["lcp"]
        assert _md5(eng.table(esa.TAB_BWT)) == BASELINE_MD["bwt"]
    assert stats["prefixlength"] == 11 and stats["maxbranchdepth"] == 27
    assert stats["largelcpvalues"] == 0
    assert "%.2f" % (stats["lcptabsum"] / (n + 1)) == "13.16"
