import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sufr_amd
def acgt(g, n): return torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")[torch.randint(0, 4, (n,), generator=g, device="cuda")]
def run(name, x, **fl):
    db = sufr_amd.DeviceBuilder(0)
    for _ in range(2): sa, lcp = db.sort(x, raw_text=True, is_dna=True, **fl)
    st = db.stats
    print(f"{name:34s} s={sa.numel():9d} levels={st.num_levels:3d} total {st.ms_total:8.2f} deep {st.ms_deep:8.2f} maxlcp {int(lcp.max())}", flush=True)
    db.close()
g = torch.Generator(device="cuda"); g.manual_seed(3)
n = 200_000_000
x = acgt(g, n + 1); x[-1] = ord("$")
rng = np.random.default_rng(5)
for hi in (5.0, 6.0, 6.7):
    y = x.clone()
    lens = (10 ** rng.uniform(2, hi, 300)).astype(np.int64)
    at = 1000
    for L in lens:
        if at + L + 1000 >= n: break
        y[at:at + int(L)] = ord("N"); at += int(L) + int(rng.integers(1000, 400_000))
    run(f"300 N runs up to 1e{hi} -a", y, allow_ambiguity=True)
