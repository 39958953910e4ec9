"""Does HBM bandwidth depend on where an allocation lands?  Streaming copy of 12 GB between freshly
allocated buffers, re-allocated every round (python profiles/placement_probe.py)."""
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
keep = []
for round_ in range(10):
    src = torch.empty(12 * GB, dtype=torch.uint8, device="cuda")
    dst = torch.empty(12 * GB, dtype=torch.uint8, device="cuda")
    src.zero_(); dst.zero_()
    t_copy = timed(lambda: dst.copy_(src))
    t_fill = timed(lambda: dst.fill_(1))
    t_read = timed(lambda: src.view(torch.int64).sum())
    print(f"round {round_}: src {src.data_ptr():#x} dst {dst.data_ptr():#x} copy {24 * GB / t_copy / 1e6:.0f} GB/s "
          f"fill {12 * GB / t_fill / 1e6:.0f} GB/s read {12 * GB / t_read / 1e6:.0f} GB/s", flush=True)
    if round_ % 3 == 2:
        keep.append(torch.empty(7 * GB, dtype=torch.uint8, device="cuda"))   # shift later placements
    del src, dst
    torch.cuda.empty_cache()
