#!/usr/bin/env python3
"""Time the engine on a device-generated synthetic sequence and run cheap
sanity checks (permutation checksum, sampled order/LCP check).  Dev tool."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genometools_amd import _lib, esa, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=float, default=16e6)
    ap.add_argument("--model", type=int, default=0)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--want", type=int, default=7)
    ap.add_argument("--ab", default="", help="env var to toggle 0/1 between runs")
    ap.add_argument("--abvals", default="0,1")
    ap.add_argument("--check", type=int, default=0, help="sampled pairs to verify on the CPU")
    a = ap.parse_args()
    n = int(a.n)
    lib = _lib.load()
    t0 = time.time()
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    _lib.check(lib.gtamd_synth_bytes(0, a.model, a.seed, n, buf.data_ptr()))
    torch.cuda.synchronize()
    print("synth %.2fs" % (time.time() - t0), flush=True)
    t0 = time.time()
    eng = esa.EsaEngine(n, synth.numofchars(a.model))
    print("create %.2fs" % (time.time() - t0), flush=True)
    eng.set_sequence_device(buf.data_ptr(), n)
    for r in range(a.runs):
        if a.ab:
            os.environ[a.ab] = a.abvals.split(',')[r % len(a.abvals.split(','))]
        t0 = time.time()
        eng.run(a.want)
        dt = time.time() - t0
        tm, st = eng.timing(), eng.stats()
        print(("[%s=%s] " % (a.ab, os.environ.get(a.ab)) if a.ab else "") + "run%d wall %.3fs dev %.1f ms -> %.3f Gbp/s | keygen %.1f sort %.1f (scatter %.1f/%d) fin %.1f refine %.1f fix %.1f | tied %d (pairs %d) rounds %d large %d maxlcp %d | msd big %d crowded %d | device memory %.1f GB" % (
            r, dt, tm["total_ms"], n / tm["total_ms"] / 1e6, tm["keygen_ms"], tm["sort_ms"],
            tm["scatter_ms"], tm["scatter_launches"], tm["finalize_ms"], tm["refine_ms"],
            tm["tie_fix_ms"], st["tied_suffixes"], st["pair_suffixes"], st["refine_rounds"],
            st["largelcpvalues"], st["maxbranchdepth"], st["msd_big_entries"], st["msd_crowded_entries"],
            st["device_bytes"] / 1e9), flush=True)
    if a.check:
        N = n + 1
        # permutation checksum on the device
        sa = torch.empty(0)
        ptr = eng.device_pointer(esa.TAB_SUF)
        # wrap the device table without copying
        class _W:  # minimal __cuda_array_interface__ holder
            pass
        w = _W()
        w.__cuda_array_interface__ = {"shape": (N,), "typestr": "<i8", "data": (ptr, False), "version": 2}
        sa = torch.as_tensor(w, device="cuda:0")
        s = int(sa.sum().item())
        assert s == N * (N - 1) // 2, "sum checksum"
        rng = np.random.default_rng(1)
        idx = np.sort(rng.integers(1, N, a.check))
        enc = buf.cpu().numpy()
        bad = 0
        for i in idx:
            pr = eng.table(esa.TAB_SUF, int(i) - 1, 2)
            lc = int(eng.table(esa.TAB_LCP, int(i), 1)[0]) if a.want & 2 else None
            p, q = int(pr[0]), int(pr[1])
            l = 0
            while p + l < n and q + l < n and enc[p + l] < 254 and enc[p + l] == enc[q + l]:
                l += 1
            ka = 256 + p + l if (p + l >= n or enc[p + l] >= 254) else int(enc[p + l])
            kb = 256 + q + l if (q + l >= n or enc[q + l] >= 254) else int(enc[q + l])
            if not (ka < kb) or (lc is not None and min(l, 255) != lc):
                bad += 1
                if bad < 5:
                    print("BAD at", i, p, q, l, lc, ka, kb)
        print("sampled check: %d pairs, %d bad" % (len(idx), bad))
    eng.close()


if __name__ == "__main__":
    main()
