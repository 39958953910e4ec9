"""gpurun -- 'python profiles/out_of_core_probe.py <n> [--budget BYTES] [--exact]': the out-of-core form of `sufr create` at full size.

A text of n bytes (random DNA, 50 kb repeats planted across the window ends and 2^32, an N stretch) is written as FASTA to /dev/shm
and built by the NATIVE CLI (`sufr create -d [--array-budget BYTES]`): the build runs shard after shard (ranges of the first 8 bytes),
every shard's slices are streamed into the one file; the whole arrays are never resident.  Without --budget the CLI first tries the
whole-array path and splits by itself when the arrays do not fit (n = 9e9: 144 GB of arrays beside ~150 GB of text and workspace).

Check: the file's SA / LCP sections are loaded back to the device and
  --exact   compared element for element with the whole-array build of the same text (DeviceBuilder, n = 4.4e9 fits), else
  (default) verified by properties: SA is a permutation of the eligible positions, 400 000 sampled adjacent ranks (100 000 of them
            with LCP >= 40) are in order with the exact LCP (tests/gpu_verify.py).
"""
import argparse
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))

import numpy as np
import torch

import sufr_amd
import gpu_verify as verify


def make_text(n: int) -> torch.Tensor:
    dev = "cuda"
    g = torch.Generator(device=dev); g.manual_seed(8)
    x = torch.empty(n, dtype=torch.uint8, device=dev)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    for lo in range(0, n, 1 << 28):
        m = min(1 << 28, n - lo)
        x[lo:lo + m] = lut[torch.randint(0, 4, (m,), generator=g, device=dev)]
    seg = x[1000:1000 + 50_000].clone()
    for at in (n // 2 - 20_000, n // 2 + 3_000_000, n - 60_000, 2_000_000_000, (1 << 32) - 25_000, n // 3 - 10_000, 2 * (n // 3) - 30_000):
        if 0 < at < n - 60_000:
            x[at:at + seg.numel()] = seg
    x[n // 2 - 5_000_000:n // 2 - 4_999_000] = ord("N")
    x[-1] = ord("$")
    return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", type=float)
    ap.add_argument("--budget", type=float, default=0)
    ap.add_argument("--exact", action="store_true")
    ap.add_argument("--devices", default="0")
    ap.add_argument("--window", type=float, default=0, help="forced window (small-scale rehearsals)")
    a = ap.parse_args()
    n = int(a.n) + 1
    shm = Path("/dev/shm")
    fa, out = shm / "ooc_probe.fa", shm / "ooc_probe.sufr"
    x = make_text(n)
    t0 = time.time()
    host = x.cpu().numpy()
    with open(fa, "wb") as f:
        f.write(b">chr\n"); f.write(memoryview(host[:-1])); f.write(b"\n")       # (the reader appends the '$')
    print(f"text: {n} bytes, FASTA written in {time.time() - t0:.1f} s", flush=True)
    want = None
    if a.exact:
        db = sufr_amd.DeviceBuilder(0)
        if a.window:
            db.ctx.set_window(int(a.window), 0)
        sa, lcp = db.sort(x, is_dna=True, index_width=8 if n >= 0xFFFFFFFF else 4)
        print(f"whole-array build on the device: {db.stats.ms_total:.0f} ms, {sa.numel()} suffixes", flush=True)
        want = (sa.cpu(), lcp.cpu())
        del sa, lcp
        db.close()
    del x
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    cmd = [str(sufr_amd.CLI_PATH), "--devices", a.devices, "create", "-d", str(fa), "-o", str(out), "--log", "debug"]
    if a.budget:
        cmd += ["--array-budget", str(int(a.budget))]
    if a.window:
        cmd += ["--window", str(int(a.window))]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, SUFR_HIP_DEBUG="1"))
    wall = time.time() - t0
    print(f"$ {' '.join(cmd)}\n  rc {r.returncode}, {wall:.1f} s wall, file {out.stat().st_size if out.exists() else 0} bytes, free HBM before {free0 / 1e9:.0f} GB", flush=True)
    lines = (r.stdout + r.stderr).strip().splitlines()
    for line in [l for l in lines if "windowed create" in l] + [l for l in lines if "[sufr_hip]" not in l][-8:]:
        print("  | " + line)
    if r.returncode != 0:
        sys.exit(1)
    f = sufr_amd.SufrFile(str(out))
    s = f.len_suffixes
    assert f.text_len == n and f.index_width == (8 if n >= 0xFFFFFFFF else 4)
    dt = torch.int64 if f.index_width == 8 else torch.int32
    need = 2 * s * f.index_width + n + (8 << 30)
    for _ in range(120):                                       # (the CLI leaves through _exit: the driver hands its HBM back a little later)
        if torch.cuda.mem_get_info()[0] >= need:
            break
        time.sleep(1.0)
    print(f"free HBM after the create: {torch.cuda.mem_get_info()[0] / 1e9:.0f} GB (the check needs {need / 1e9:.0f})", flush=True)
    t0 = time.time()
    d_sa = torch.empty(s, dtype=dt, device="cuda"); d_lcp = torch.empty(s, dtype=dt, device="cuda")
    fsa, flcp = f.suffix_array, f.lcp
    sdt = np.int64 if f.index_width == 8 else np.int32
    for lo in range(0, s, 1 << 28):
        hi = min(s, lo + (1 << 28))
        d_sa[lo:hi] = torch.from_numpy(fsa[lo:hi].view(sdt).copy()).cuda()
        d_lcp[lo:hi] = torch.from_numpy(flcp[lo:hi].view(sdt).copy()).cuda()
    ftext = f.text
    d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
    for lo in range(0, n, 1 << 28):                          # (chunks: elementwise torch ops on > 2^32 elements are not to be trusted)
        hi = min(n, lo + (1 << 28))
        d_text[lo:hi] = torch.from_numpy(ftext[lo:hi].copy()).cuda()
        ne = (d_text[lo:hi] != torch.from_numpy(host[lo:hi]).cuda()).nonzero()
        assert ne.numel() == 0, f"text section differs from the input at {ne.numel()} positions of [{lo}, {hi}), first {ne[:4, 0].tolist()}"
    del ftext
    print(f"file sections back on the device in {time.time() - t0:.1f} s; text section = the input", flush=True)
    if want is not None:
        assert s == want[0].numel()
        for lo in range(0, s, 1 << 28):
            hi = min(s, lo + (1 << 28))
            assert torch.equal(d_sa[lo:hi].cpu(), want[0][lo:hi]), f"SA differs in [{lo}, {hi})"
            assert torch.equal(d_lcp[lo:hi].cpu(), want[1][lo:hi]), f"LCP differs in [{lo}, {hi})"
        print(f"EQUAL: SA and LCP of the out-of-core file = the whole-array build, {s} suffixes, element for element")
    else:
        count = verify.check_permutation(d_text, d_sa, is_dna=True, raw_is_normalised=True)
        assert count == s, (count, s)
        res = verify.check_sampled_ranks(d_text, d_sa, d_lcp, samples=400_000, deep_samples=100_000, deep_min_lcp=40)
        assert int(d_sa.max()) > (1 << 32) or n < (1 << 32)
        print(f"VERIFIED: SA is a permutation of the {count} eligible positions; sampled ranks {res}")
    del fsa, flcp
    f.close()
    for p in (fa, out):
        p.unlink(missing_ok=True)


if __name__ == "__main__":
    main()
