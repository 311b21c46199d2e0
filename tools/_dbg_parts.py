import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_util as ou
from thread_comm import build_in_parts
rng = np.random.default_rng(3)
a = rng.integers(0, 4, 3000, dtype=np.uint8)
cases = [np.zeros(4000, dtype=np.uint8), np.full(500, 254, dtype=np.uint8),
         np.concatenate([a, [255], a, [254], a]).astype(np.uint8),
         np.tile(np.array([0, 1, 2], dtype=np.uint8), 2000), a[:5]]
for ci, enc in enumerate(cases):
    for parts in (4, 8):
        tabs, stats, per = build_in_parts(enc, 4, parts)
        ora = ou.esa(enc, 4)
        for k in ("suf", "bwt", "lcp", "llv"):
            if not np.array_equal(tabs[k], ora[k]):
                d = np.nonzero(tabs[k].reshape(-1)[:ora[k].size] != ora[k].reshape(-1)[:tabs[k].size])[0] if tabs[k].size == ora[k].size else []
                print("case", ci, "parts", parts, k, "differs at", len(d), "places", d[:10], "rounds", stats["refine_rounds"], flush=True)
print("done")
