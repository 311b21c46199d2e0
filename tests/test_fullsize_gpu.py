"""BASELINE.json sizes, checked through size-independent properties (the CPU
oracle's comparison sort cannot run at these sizes):

* configs[1], 256 Mbp uniform DNA: the suffix array passes the linear-time
  checker (permutation + sortedness, oracle/esa_oracle.c ora_check_suffix_array)
  and LCP/BWT equal the tables the oracle derives from that suffix array
  (Kasai).  Together that is bit-exactness of all three tables.
* configs[2], 3 Gbp human-like DNA: permutation checksum over the whole suffix
  array on the device, plus order and LCP of thousands of sampled neighbour
  pairs re-derived on the CPU from the encoded sequence, plus the tail layout
  (specials in text order, then n).
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


def test_config2_3gbp_humanlike_properties(gpu):
    n = 3 * 1000 * 1000 * 1000
    N = n + 1
    buf = _device_sequence(synth.MODEL_HUMANLIKE_DNA, 43, n)
    with esa.EsaEngine(n, 4) as eng:
        eng.set_sequence_device(buf.data_ptr(), n)
        eng.run()
        st = eng.stats()
        # permutation: sum and sum of squares (mod 2^64) of the whole table
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
        # tail: every suffix that starts with a special, in text order, then n
        tail = eng.table(esa.TAB_SUF, N - 1 - 2000, 2001)
        assert tail[-1] == n
        assert np.all(enc[tail[:-1].astype(np.int64)] >= 254)
        assert np.all(np.diff(tail[:-1].astype(np.int64)) > 0)
        first_special = eng.table(esa.TAB_SUF, N - 1 - specials, 1)[0]
        assert enc[int(first_special)] >= 254
        before_tail = eng.table(esa.TAB_SUF, N - 2 - specials, 1)[0]
        assert enc[int(before_tail)] < 254
        # sampled neighbour pairs: order, LCP byte, BWT byte
        rng = np.random.default_rng(2026)
        idx = np.sort(rng.integers(1, N - specials - 1, 4000))
        bad = 0
        for i in idx:
            i = int(i)
            p, q = (int(x) for x in eng.table(esa.TAB_SUF, i - 1, 2))
            lcpb = int(eng.table(esa.TAB_LCP, i, 1)[0])
            bwtb = int(eng.table(esa.TAB_BWT, i, 1)[0])
            l = 0
            while p + l < n and q + l < n and enc[p + l] < 254 and enc[p + l] == enc[q + l]:
                l += 1
            ka = 256 + p + l if (p + l >= n or enc[p + l] >= 254) else int(enc[p + l])
            kb = 256 + q + l if (q + l >= n or enc[q + l] >= 254) else int(enc[q + l])
            ok = ka < kb and lcpb == min(l, 255) and bwtb == (254 if q == 0 else int(enc[q - 1]))
            bad += not ok
        assert bad == 0
        # every .llv entry: index ascending, value >= 255, lcp byte is 255
        llv = eng.table(esa.TAB_LLV)
        assert len(llv) == st["largelcpvalues"] > 0
        assert np.all(np.diff(llv[:, 0].astype(np.int64)) > 0)
        assert int(llv[:, 1].min()) >= 255 and int(llv[:, 1].max()) == st["maxbranchdepth"]
        some = llv[rng.integers(0, len(llv), 200)]
        for i, v in some:
            assert int(eng.table(esa.TAB_LCP, int(i), 1)[0]) == 255
            p, q = (int(x) for x in eng.table(esa.TAB_SUF, int(i) - 1, 2))
            v = int(v)
            assert np.array_equal(enc[p:p + v], enc[q:q + v]) and np.all(enc[p:p + v] < 254)
            assert p + v >= n or q + v >= n or enc[p + v] >= 254 or enc[q + v] >= 254 or enc[p + v] != enc[q + v]
