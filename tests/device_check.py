"""Exact checks of tables that are too large for the CPU oracle, with torch on the
device that holds them.  TEST INFRASTRUCTURE (the checker, never the thing
checked): a restatement of the reference's own linear-time checkers

  gt_suftab_lightweightcheck   src/match/sfx-lwcheck.c:181-337
  gt_lcptab_lightweightcheck   src/match/sfx-linlcp.c:548

for tables in device memory.  Sortedness of a suffix array is a LOCAL property
once it is a permutation: with rank = its inverse,

    suffix SA[i-1] < suffix SA[i]   for all i
  <=>  for all i:  c(SA[i-1]) < c(SA[i])  or
                   c(SA[i-1]) = c(SA[i]) is a letter and rank[SA[i-1]+1] < rank[SA[i]+1]

(induction over the common prefix), where c(p) is the letter at p, or -- special
symbols being unique and larger than every letter, larger at larger positions
(src/core/encseq.h:640, src/match/sfx-bentsedg.c:75-80) -- 256 + p.
"""
import numpy as np
import torch

CHUNK = 1 << 27


def _chunks(n, step=CHUNK):
    for a in range(0, n, step):
        yield a, min(n, a + step)


def check_suffix_array_exact(sa, enc):
    """sa: int64 device tensor, N = n + 1 entries; enc: uint8 device tensor, n
    encoded symbols.  Returns (ok, message)."""
    N = sa.numel()
    n = N - 1
    dev = sa.device
    # ---- permutation: the inverse exists and inverts
    rank = torch.empty(N, dtype=torch.int64, device=dev)
    rank.fill_(-1)
    for a, b in _chunks(N):
        p = sa[a:b]
        if int(p.min()) < 0 or int(p.max()) > n:
            return False, "entry outside [0, n] in [%d, %d)" % (a, b)
        rank[p] = torch.arange(a, b, dtype=torch.int64, device=dev)
    for a, b in _chunks(N):
        if not bool((sa[rank[a:b]] == torch.arange(a, b, dtype=torch.int64, device=dev)).all()):
            return False, "not a permutation (positions [%d, %d))" % (a, b)
    # ---- order of every pair of neighbours
    def c_of(p):
        sym = torch.where(p < n, enc[torch.clamp(p, max=n - 1)].to(torch.int64),
                          torch.full_like(p, 255))
        return torch.where(sym >= 254, 256 + p, sym)
    for a, b in _chunks(N - 1):
        p, q = sa[a:b], sa[a + 1:b + 1]
        cp, cq = c_of(p), c_of(q)
        rp = rank[torch.clamp(p + 1, max=n)]
        rq = rank[torch.clamp(q + 1, max=n)]
        ok = (cp < cq) | ((cp == cq) & (cp < 254) & (rp < rq))
        if not bool(ok.all()):
            i = int(torch.nonzero(~ok)[0].item()) + a + 1
            return False, "suffixes out of order at table index %d" % i
    del rank
    return True, ""


def lcp_of_pairs(enc, p, q, cap=None):
    """longest common prefix of the suffixes p[k], q[k] (int64 device tensors):
    equal letters count, a special on either side ends the match"""
    n = enc.numel()
    m = p.numel()
    l = torch.zeros(m, dtype=torch.int64, device=p.device)
    active = torch.arange(m, dtype=torch.int64, device=p.device)
    step = 0
    while active.numel() > 0:
        pa, qa, la = p[active], q[active], l[active]
        # 16 symbols per sweep: the first mismatch inside them
        off = torch.arange(16, dtype=torch.int64, device=p.device)
        ia = pa[:, None] + la[:, None] + off[None, :]
        ib = qa[:, None] + la[:, None] + off[None, :]
        va = torch.where(ia < n, enc[torch.clamp(ia, max=n - 1)], torch.full_like(ia, 255, dtype=torch.uint8))
        vb = torch.where(ib < n, enc[torch.clamp(ib, max=n - 1)], torch.full_like(ib, 255, dtype=torch.uint8))
        stop = (va != vb) | (va >= 254)
        first = torch.where(stop.any(dim=1), stop.to(torch.int8).argmax(dim=1),
                            torch.full((active.numel(),), 16, dtype=torch.int64, device=p.device))
        l[active] = la + first
        keep = first == 16
        if cap is not None:
            keep &= (la + first) < cap
        active = active[keep]
        step += 1
    return l


def check_bwt_exact(sa, enc, bwt):
    N = sa.numel()
    for a, b in _chunks(N):
        p = sa[a:b]
        want = torch.where(p > 0, enc[torch.clamp(p - 1, min=0)], torch.full_like(p, 254, dtype=torch.uint8))
        if not bool((want == bwt[a:b]).all()):
            i = int(torch.nonzero(want != bwt[a:b])[0].item()) + a
            return False, "bwt differs at table index %d" % i
    return True, ""


def check_lcp_samples(sa, enc, lcp, llv_idx, llv_val, idx):
    """idx: int64 device tensor of table indices >= 1; the LCP byte of each, and
    for bytes of 255 the entry of .llv, against the symbols themselves"""
    p, q = sa[idx - 1], sa[idx]
    l = lcp_of_pairs(enc, p, q)
    byte = lcp[idx].to(torch.int64)
    if not bool((byte == torch.clamp(l, max=255)).all()):
        k = int(torch.nonzero(byte != torch.clamp(l, max=255))[0].item())
        return False, "lcp byte at table index %d is %d, the suffixes share %d" % (
            int(idx[k]), int(byte[k]), int(l[k]))
    big = l >= 255
    if bool(big.any()):
        bi = idx[big]
        pos = torch.searchsorted(llv_idx, bi)
        if int(pos.max()) >= llv_idx.numel() or not bool((llv_idx[pos] == bi).all()):
            return False, "an lcp >= 255 has no .llv entry"
        if not bool((llv_val[pos] == l[big]).all()):
            return False, "a .llv value differs from the symbols"
    return True, ""


def check_llv_all(sa, enc, lcp, llv_idx, llv_val, rng_seed=1, probes=16):
    """EVERY .llv entry: index ascending, byte 255 in the table, the suffixes
    differ (or one ends / meets a special) exactly at offset value, agree at value-1
    and at `probes` random offsets below"""
    m = llv_idx.numel()
    n = enc.numel()
    if m == 0:
        return True, ""
    if not bool((llv_idx[1:] > llv_idx[:-1]).all()):
        return False, ".llv indices not ascending"
    g = torch.Generator(device=sa.device)
    g.manual_seed(rng_seed)
    for a, b in _chunks(m, 1 << 24):
        i, v = llv_idx[a:b], llv_val[a:b]
        if not bool((lcp[i] == 255).all()) or int(v.min()) < 255:
            return False, ".llv entry without a 255 in the lcp table, or value < 255"
        p, q = sa[i - 1], sa[i]
        ia, ib = p + v, q + v
        va = torch.where(ia < n, enc[torch.clamp(ia, max=n - 1)], torch.full_like(ia, 255, dtype=torch.uint8))
        vb = torch.where(ib < n, enc[torch.clamp(ib, max=n - 1)], torch.full_like(ib, 255, dtype=torch.uint8))
        if not bool(((va != vb) | (va >= 254)).all()):
            return False, "suffixes of an .llv entry agree beyond its value"
        r = (torch.rand((b - a, probes), device=sa.device, generator=g) * v[:, None].to(torch.float64)).to(torch.int64)
        r = torch.cat([r, (v - 1)[:, None]], dim=1)
        xa, xb = enc[p[:, None] + r], enc[q[:, None] + r]
        if not bool(((xa == xb) & (xa < 254)).all()):
            return False, "suffixes of an .llv entry differ before its value"
    return True, ""


def count_lcp_overflows(lcp):
    total = 0
    for a, b in _chunks(lcp.numel()):
        total += int((lcp[a:b] == 255).sum().item())
    return total


def as_tensor(ptr, count, typestr, device="cuda:0"):
    class _W:
        pass
    w = _W()
    w.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2}
    return torch.as_tensor(w, device=device)


def numpy_pairs_to_device(llv, device="cuda:0"):
    t = torch.from_numpy(np.ascontiguousarray(llv).view(np.int64).reshape(-1, 2)).to(device)
    return t[:, 0].contiguous(), t[:, 1].contiguous()
