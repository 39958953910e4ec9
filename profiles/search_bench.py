"""Rate of the batched device search (k_search_batch):   python profiles/search_bench.py [text_len] [num_queries] [query_len]
Builds the index of a synthetic genome on the device, wraps it in place, searches `num_queries` substrings of the text
(every tenth with one changed symbol) and prints queries/s from HIP events around the launch; then the same queries
through the host search of a written file when the text is small enough to write (<= 200 Mb)."""
import ctypes as C
import os
import sys
import tempfile
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sufr_amd
from sufr_amd import synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
ql = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = "cuda"
x, _ = synth.syn_human(n, seed=4, device=dev)
db = sufr_amd.DeviceBuilder(0)
lib = sufr_amd.lib()
# the index holds the normalized text: normalize in place the way the file would hold it (soft-mask kept: --dna only)
norm = torch.where((x >= 97) & (x <= 122), x - 32, x)        # sufr_builder.rs:144-160 without --ignore-softmask: upper-case
del x
sa, lcp = db.sort(norm, is_dna=True)
del lcp
ix = sufr_amd.DeviceIndex.wrap(db.ctx, norm, sa, is_dna=True, prefix_table=not os.environ.get("SEARCH_NO_TABLE"))
g = torch.Generator(device=dev); g.manual_seed(1)
at = torch.randint(0, n - ql - 1, (nq,), generator=g, device=dev)
qb = norm[(at[:, None] + torch.arange(ql, device=dev)[None, :]).reshape(-1)].contiguous()
flip = torch.arange(0, nq, 10, device=dev) * ql + torch.randint(0, ql, ((nq + 9) // 10,), generator=g, device=dev)
qb[flip] = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (flip.numel(),), generator=g, device=dev)]
off = (torch.arange(nq + 1, device=dev, dtype=torch.int64) * ql).contiguous()
stream = torch.cuda.Stream()
lib.sufr_hip_set_stream(db.ctx.handle, stream.cuda_stream)
with torch.cuda.stream(stream):
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        lo, hi = ix.search_device(qb, off, wait=False)
        e1.record(stream)
        stream.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"device{' (no prefix table)' if os.environ.get('SEARCH_NO_TABLE') else ''}: text {n:,} suffixes {sa.numel():,} queries {nq:,} x {ql}: {ms:.2f} ms  {nq / ms / 1e3:.1f} M queries/s  "
              f"found {(hi > lo).float().mean().item():.3f}  mean count {(hi - lo).float().mean().item():.2f}", flush=True)
if n <= 200_000_000:
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "x.sufr")
        text_h = norm.cpu().numpy(); sa_h = sa.cpu().numpy().view(np.uint32); lcp_h = np.zeros_like(sa_h)
        starts = np.zeros(1, dtype=np.uint64); names = (C.c_char_p * 1)(b"1"); err = C.create_string_buffer(256)
        assert lib.sufr_write_file(path.encode(), 1, 0, 0, text_h.ctypes.data, n, 4, sa_h.ctypes.data, lcp_h.ctypes.data, sa_h.size,
                                   0, 0, None, starts.ctypes.data, 1, names, err, len(err)) == 0
        f = sufr_amd.SufrFile(path)
        m = min(nq, 2_000_000)
        qh = qb[:m * ql].cpu().numpy()
        offs = (np.arange(m + 1, dtype=np.uint64) * ql)
        lo_h = lo[:m].cpu().numpy().astype(np.uint64); hi_h = hi[:m].cpu().numpy().astype(np.uint64)
        a = np.zeros(m, dtype=np.uint64); b = np.zeros(m, dtype=np.uint64)
        for threads in (1, 16, os.cpu_count() or 1):
            mm = m if threads > 1 else m // 10
            t0 = time.perf_counter()
            lib.sufr_file_search_batch(f._h, qh.ctypes.data, offs.ctypes.data, mm, 0, 0, a.ctypes.data, b.ctypes.data, threads)
            dt = time.perf_counter() - t0
            assert np.array_equal(a[:mm], lo_h[:mm]) and np.array_equal(b[:mm], hi_h[:mm])
            print(f"host (sufr_file_search_batch, {threads} threads): {mm:,} queries in {dt:.2f} s  {mm / dt / 1e6:.3f} M queries/s, equal to the device answers", flush=True)
