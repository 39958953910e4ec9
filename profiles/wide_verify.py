"""Windowed build at full size with a chosen window, checked by properties:   python profiles/wide_verify.py <text_len> <window> <margin>
(the three-window path and the margin retry at a size no second build can confirm)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sufr_amd
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import gpu_verify as verify

n = int(float(sys.argv[1])); window = int(float(sys.argv[2])); margin = int(float(sys.argv[3]))
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(11)
x = torch.empty(n, dtype=torch.uint8, device=dev)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
for lo in range(0, n, 1 << 28):
    m = min(1 << 28, n - lo)
    x[lo:lo + m] = lut[torch.randint(0, 4, (m,), generator=g, device=dev)]
seg = x[1000:41000].clone()
for at in (window - 15_000, 2 * window - 100, n - 50_000, n // 5, (1 << 32) - 20_000):
    if 0 < at < n - seg.numel():
        x[at:at + seg.numel()] = seg
x[-1] = ord("$")
db = sufr_amd.DeviceBuilder(0)
db.ctx.set_window(window, margin)
sa, lcp = db.sort(x, is_dna=True, index_width=8)
st = db.stats
print(f"n={n:,} window={window:,} margin={margin:,}: {sa.numel():,} suffixes, device total {st.ms_total:.0f} ms", flush=True)
cnt = verify.check_permutation(x, sa, is_dna=True, raw_is_normalised=True)
res = verify.check_sampled_ranks(x, sa, lcp, samples=400_000, deep_samples=100_000, deep_min_lcp=40)
print("permutation of", cnt, "suffix starts;", res, flush=True)
