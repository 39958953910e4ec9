"""Does HBM bandwidth depend on where an allocation lands?  Streaming copy of 12 GB between freshly
allocated buffers, re-allocated every round, with the destination shifted by a few offsets inside its
allocation (python profiles/placement_probe.py)."""
import torch

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best

GB = 1 << 30
N = 12 * GB
SKEWS = [0, 4 << 10, 64 << 10, 256 << 10, (1 << 20) + (4 << 10), (2 << 20), (3 << 20) + (192 << 10), 16 << 20, (64 << 20) + (68 << 10)]
for round_ in range(8):
    src = torch.empty(N + (128 << 20), dtype=torch.uint8, device="cuda")
    dst = torch.empty(N + (128 << 20), dtype=torch.uint8, device="cuda")
    src.zero_(); dst.zero_()
    res = []
    for sk in SKEWS:
        d = dst[sk:sk + N]
        t = timed(lambda: d.copy_(src[:N]))
        res.append(2 * N / t / 1e6)
    t_fill = timed(lambda: dst[:N].fill_(1))
    t_read = timed(lambda: src[:N].view(torch.int64).sum())
    print(f"round {round_}: copy GB/s by dst skew " + " ".join(f"{r:.0f}" for r in res) +
          f" | fill {N / t_fill / 1e6:.0f} read {N / t_read / 1e6:.0f}", flush=True)
    del src, dst, d
    torch.cuda.empty_cache()
