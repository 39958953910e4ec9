"""Windowed build of a text beyond 32-bit indices:   python profiles/wide_bench.py [text_len] [window] [margin]
Random DNA with a few planted 50 kb repeats; prints the device time of the second build."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sufr_amd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_400_000_001
window = int(float(sys.argv[2])) if len(sys.argv) > 2 else 0
margin = int(float(sys.argv[3])) if len(sys.argv) > 3 else 0
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(8)
x = torch.empty(n, dtype=torch.uint8, device=dev)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
for lo in range(0, n, 1 << 28):
    m = min(1 << 28, n - lo)
    x[lo:lo + m] = lut[torch.randint(0, 4, (m,), generator=g, device=dev)]
seg = x[1000:51000].clone()
for at in (n // 2 - 20_000, n // 2 + 3_000_000, n - 60_000, n // 3):
    x[at:at + seg.numel()] = seg
x[-1] = ord("$")
db = sufr_amd.DeviceBuilder(0)
db.ctx.set_window(window, margin)
sa = torch.empty(n, dtype=torch.int64, device=dev)
lcp = torch.empty(n, dtype=torch.int64, device=dev)
for rep in range(2):
    s, l = db.sort(x, is_dna=True, index_width=8, out_sa=sa, out_lcp=lcp)
    st = db.stats
    print(f"n={n:,} s={s.numel():,} window={window} margin={margin}: device total {st.ms_total:.1f} ms "
          f"({s.numel() / st.ms_total / 1e6:.3f} G suffixes/s)", flush=True)
