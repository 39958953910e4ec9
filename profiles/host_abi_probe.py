"""The host-buffer ABI (sufr_hip_build_u32: what the libsufr shim of INTEGRATION.md section 3 binds) on the 3.1 Gb stand-in:
pageable host text in, normalised text + SA + LCP out into pageable host arrays.   python profiles/host_abi_probe.py [bases] [reps]
Prints seconds per call (fresh destination arrays every call: their first-touch faults are part of the call, as they are for a
caller that has just allocated them) and the phases the library reports in sufr_hip_stats (host_read_s = H2D, host_build_s,
host_write_s = D2H)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
import sufr_amd
from sufr_amd import synth, _lib

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
x, _ = synth.syn_human(bases, seed=4, device="cuda")
raw = x.cpu().numpy()
del x
torch.cuda.empty_cache()
n = raw.size
ctx = sufr_amd.Context(0)
L = _lib.lib()
flags = _lib.FLAG_DNA | _lib.FLAG_IGNORE_SOFTMASK | _lib.FLAG_RAW_TEXT
for rep in range(reps):
    norm = np.empty(n, dtype=np.uint8)
    sa = np.empty(n, dtype=np.uint32)          # (capacity n: the caller does not know s)
    lcp = np.empty(n, dtype=np.uint32)
    ns = C.c_uint64(0)
    st = _lib.Stats()
    t0 = time.perf_counter()
    rc = L.sufr_hip_build_u32(ctx.handle, raw.ctypes.data, n, flags, 0, None, 256, 42, norm.ctypes.data, sa.ctypes.data, lcp.ctypes.data,
                              n, C.byref(ns), C.byref(st))
    dt = time.perf_counter() - t0
    ctx.check(rc)
    if rep == 0:        # the text that came back under the build is the reference's text map of the input (lowercase -> 'N')
        want = np.where((raw >= 97) & (raw <= 122), np.uint8(78), raw)
        assert np.array_equal(norm, want), "normalised text differs"
        assert int(sa[:ns.value].astype(np.uint64).sum()) > 0 and int(lcp[0]) == 0
        del want
    gb = (2 * n + 8 * ns.value) / 1e9
    print(f"call {rep}: {dt:.3f} s  ({gb:.1f} GB over PCIe: {gb / dt:.1f} GB/s)  H2D {st.host_read_s:.3f} s, build {st.host_build_s:.3f} s "
          f"(device {st.ms_total:.1f} ms), D2H {st.host_write_s:.3f} s  suffixes={ns.value}", flush=True)
    del norm, sa, lcp
ctx.close()
