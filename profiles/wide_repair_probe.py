import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import sufr_amd
from oracle_helper import Oracle
from test_gpu_wide import repeat_text, build_capped
o = Oracle()
t = repeat_text(60_000, 1, 900, 40)
for flags in (dict(is_dna=True), dict()):
    want_sa, want_lcp, _ = o.build(t, **flags)
    for (w, m, r) in [(5000, 300, 600), (4096, 64, 64), (8000, 100, 500), (20000, 16, 16), (5000, 300, 100)]:
        try:
            sa, lcp, rep = build_capped(torch.from_numpy(t).cuda(), w, m, r, 4, **flags)
            print(flags, w, m, r, "repaired", rep, "sa_eq", np.array_equal(sa, want_sa.astype(np.uint64)), "lcp_eq", np.array_equal(lcp, want_lcp.astype(np.uint64)),
                  "ndiff", int((sa != want_sa.astype(np.uint64)).sum()) if sa.size == want_sa.size else -1, flush=True)
        except Exception as e:
            print(flags, w, m, r, "ERR", e, flush=True)
rng = np.random.default_rng(11)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
t = acgt[rng.integers(0, 4, 50_000)].copy()
seg = t[2_000:8_000].copy()
for at in (13_500, 30_100):
    t[at:at + seg.size] = seg
t[41_000:44_000] = ord("T")
t[-1] = ord("$")
want_sa, want_lcp, _ = o.build(t, is_dna=True)
print("max lcp", want_lcp.max())
for (w, m, r) in [(7000, 200, 1000), (7000, 200, 200), (3000, 100, 100)]:
    os.environ["SUFR_HIP_DEBUG"] = "1"
    sa, lcp, rep = build_capped(torch.from_numpy(t).cuda(), w, m, r, 4, is_dna=True)
    print("dups", w, m, r, "repaired", rep, "sa_eq", np.array_equal(sa, want_sa.astype(np.uint64)), "lcp_eq", np.array_equal(lcp, want_lcp.astype(np.uint64)), flush=True)
