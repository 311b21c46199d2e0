#!/usr/bin/env python3
"""Exact check of a full-size build on the device that holds it (tests/device_check.py: the
reference's lightweight checkers restated for device tensors): the suffix table exactly,
.bwt for every entry, every .llv entry, .lcp on samples.  Dev tool for sizes and models the
test suite does not run:

  python tools/exact_probe.py --n 3000000129 --model 3 --seed 43
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import device_check as dc  # noqa: E402
from genometools_amd import _lib, esa, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=float, default=3e9)
    ap.add_argument("--model", type=int, default=synth.MODEL_HUMANLIKE_DNA)
    ap.add_argument("--seed", type=int, default=43)
    ap.add_argument("--fuzz-seconds", type=float, default=0,
                    help="instead of one build: random sizes (2^25 .. 3.1 G symbols), models, seeds and "
                         "engine switches for this long, every build checked exactly")
    ap.add_argument("--fuzz-min", type=float, default=2 ** 25, help="smallest size of the random builds")
    a = ap.parse_args()
    if a.fuzz_seconds > 0:
        return fuzz(a.fuzz_seconds, a.seed, int(a.fuzz_min))
    return check_one(int(a.n), a.model, a.seed)


def fuzz(seconds, seed, nmin=2 ** 25):
    import numpy as np
    rng = np.random.default_rng(seed)
    t0, cases = time.time(), 0
    switches = {"GTAMD_RANK_ALL_WINDOWS": ["0", "0", "1"], "GTAMD_WIN_FILTER_LDS": ["1", "1", "0"],
                "GTAMD_PAIR_CHUNK": ["16", "32", "128"], "GTAMD_APPLY_EARLY": ["0", "1", "2"],
                "GTAMD_MSD_PACK": ["0", "1"], "GTAMD_ROUND_STRIDE": ["1024", "1536", "2048"],
                "GTAMD_NO_SMALL_GROUPS": ["0", "0", "1"], "GTAMD_PAIR_LONG": ["0", "1"],
                "GTAMD_STABLE_PARTITION": ["0", "1"], "GTAMD_MSD_BIN_LIMIT": ["32", "128", "512"]}
    while time.time() - t0 < seconds:
        model = int(rng.choice([0, 1, 1, 2, 3]))
        top = 1.2e9 if model == 2 else 3.1e9
        n = int(np.exp(rng.uniform(np.log(min(nmin, top / 2)), np.log(top))))
        env = {k: str(rng.choice(v)) for k, v in switches.items()}
        os.environ.update(env)
        print("case %d: model %d n %d seed %d %s" % (cases, model, n, int(seed) + cases, env), flush=True)
        check_one(n, model, int(seed) + cases)
        torch.cuda.empty_cache()          # (the next engine allocates outside torch)
        cases += 1
    print("scale fuzz ok: %d builds in %.0f s" % (cases, time.time() - t0))
    return 0


def check_one(n, model, seed):
    class A:
        pass
    a = A()
    a.model, a.seed = model, seed
    N = n + 1
    lib = _lib.load()
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    _lib.check(lib.gtamd_synth_bytes(0, a.model, a.seed, n, buf.data_ptr()))
    torch.cuda.synchronize()
    with esa.EsaEngine(n, synth.numofchars(a.model)) as eng:
        eng.set_sequence_device(buf.data_ptr(), n)
        t0 = time.time()
        eng.run()
        st, tm = eng.stats(), eng.timing()
        print("built in %.1f ms: tied %d (pairs %d), rounds %d, large %d, maxlcp %d, rank entries built %d of %d" % (
            tm["total_ms"], st["tied_suffixes"], st["pair_suffixes"], st["refine_rounds"],
            st["largelcpvalues"], st["maxbranchdepth"], st["rank_entries_built"], N), flush=True)
        sa = dc.as_tensor(eng.device_pointer(esa.TAB_SUF), N, "<i8")
        lcp = dc.as_tensor(eng.device_pointer(esa.TAB_LCP), N, "|u1")
        bwt = dc.as_tensor(eng.device_pointer(esa.TAB_BWT), N, "|u1")
        ok, msg = dc.check_suffix_array_exact(sa, buf)
        print("suffix table exact:", ok, msg, flush=True)
        assert ok
        ok, msg = dc.check_bwt_exact(sa, buf, bwt)
        print("bwt exact:", ok, msg, flush=True)
        assert ok
        nl = eng.entries(esa.TAB_LLV)
        assert nl == st["largelcpvalues"]
        if nl:
            llv = dc.as_tensor(eng.device_pointer(esa.TAB_LLV), 2 * nl, "<i8").view(-1, 2)
            llv_idx, llv_val = llv[:, 0].contiguous(), llv[:, 1].contiguous()
            assert dc.count_lcp_overflows(lcp) == nl
            assert int(llv_val.max().item()) == st["maxbranchdepth"]
            ok, msg = dc.check_llv_all(sa, buf, lcp, llv_idx, llv_val)
            print("every .llv entry:", ok, msg, flush=True)
            assert ok
            specials = int((buf >= 254).sum().item())
            g = torch.Generator(device="cuda:0")
            g.manual_seed(7)
            idx = torch.randint(1, N - specials, (8_000_000,), device="cuda:0", generator=g)
            for b0 in range(0, idx.numel(), 1 << 22):
                ok, msg = dc.check_lcp_samples(sa, buf, lcp, llv_idx, llv_val, idx[b0:b0 + (1 << 22)])
                assert ok, msg
            print("lcp on 8 M samples: True", flush=True)
        print("checked in %.0f s" % (time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
