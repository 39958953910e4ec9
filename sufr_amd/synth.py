"""Deterministic synthetic stand-ins for the BASELINE.json genomes.

There is no network and the real assemblies are not in the reference repository, so the benchmark
and the large parity tests run on seeded synthetic texts whose statistics follow the named genomes:
iid base composition, planted repeat families, segmental duplications, tandem repeats, soft-masked
(lowercase) runs and N runs.  Generators are vectorised torch code so that the 3.1 Gb human-sized
text can be produced directly in HBM in seconds; every scatter writes disjoint positions, so a given
(seed, device type) always yields the same bytes.  Texts follow the read_sequence_file layout
(util.rs:51-89): s1 + '%' + s2 + ... + '$'."""
from __future__ import annotations

import math

import numpy as np
import torch

_CHUNK = 1 << 27


def _gen(device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def _lut(device):
    return torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)  # A C G T


def _bases(g, n: int, at_frac: float, device) -> torch.Tensor:
    """n iid bases, P(A)=P(T)=at_frac/2, P(C)=P(G)=(1-at_frac)/2."""
    out = torch.empty(n, dtype=torch.uint8, device=device)
    t1, t2, t3 = at_frac / 2, 0.5, 1 - at_frac / 2
    lut = _lut(device)
    for lo in range(0, n, _CHUNK):
        m = min(_CHUNK, n - lo)
        u = torch.rand(m, generator=g, device=device)
        idx = (u >= t1).to(torch.int64) + (u >= t2) + (u >= t3)
        out[lo:lo + m] = lut[idx]
    return out


def _mutated_copies(g, fam: torch.Tensor, copies: int, rate_lo: float, rate_hi: float, device) -> torch.Tensor:
    """copies x len(fam) matrix of the family with per-copy substitution rate in [rate_lo, rate_hi]."""
    L = fam.numel()
    rate = torch.rand(copies, 1, generator=g, device=device) * (rate_hi - rate_lo) + rate_lo
    hit = torch.rand(copies, L, generator=g, device=device) < rate
    rnd = _lut(device)[torch.randint(0, 4, (copies, L), generator=g, device=device)]
    return torch.where(hit, rnd, fam.unsqueeze(0).expand(copies, L))


def _plant_family(g, text: torch.Tensor, fam: torch.Tensor, copies: int, rate_lo: float, rate_hi: float,
                  lowercase: bool):
    """Write `copies` mutated copies of `fam` at distinct slots of a grid (no two copies overlap)."""
    n, L = text.numel(), fam.numel()
    slots = n // L
    copies = min(copies, slots)
    if copies <= 0:
        return
    dev = text.device
    step = max(1, (1 << 26) // L)
    perm = torch.randperm(slots, generator=g, device=dev)[:copies]
    ar = torch.arange(L, device=dev)
    for lo in range(0, copies, step):
        at = perm[lo:lo + step] * L
        vals = _mutated_copies(g, fam, at.numel(), rate_lo, rate_hi, dev)
        if lowercase:
            vals = vals | 0x20
        text[(at.unsqueeze(1) + ar.unsqueeze(0)).reshape(-1)] = vals.reshape(-1)


def _plant_tandem(g, text: torch.Tensor, copies: int, length: int, lowercase_frac: float):
    """`copies` tandem arrays of `length` bases, unit length 1..6, on distinct grid slots."""
    n = text.numel()
    slots = n // length
    copies = min(copies, slots)
    if copies <= 0:
        return
    dev = text.device
    perm = torch.randperm(slots, generator=g, device=dev)[:copies] * length
    unit = torch.randint(1, 7, (copies, 1), generator=g, device=dev)
    pat = _lut(dev)[torch.randint(0, 4, (copies, 6), generator=g, device=dev)]
    j = torch.arange(length, device=dev).unsqueeze(0) % unit
    vals = torch.gather(pat, 1, j)
    low = (torch.rand(copies, 1, generator=g, device=dev) < lowercase_frac).to(torch.uint8) * 0x20
    vals = vals | low
    text[(perm.unsqueeze(1) + torch.arange(length, device=dev).unsqueeze(0)).reshape(-1)] = vals.reshape(-1)


def _geometric(g, k: int, mean: float, device) -> torch.Tensor:
    u = torch.rand(k, generator=g, device=device, dtype=torch.float64).clamp_(1e-12, 1 - 1e-12)
    return (torch.log(u) / math.log(1 - 1.0 / max(mean, 1.0001))).floor().to(torch.int64) + 1


def _mask_runs(g, text: torch.Tensor, frac: float, mean_run: float):
    """Lowercase ~frac of the text in geometric runs separated by geometric gaps."""
    n = text.numel()
    if frac <= 0 or n == 0:
        return
    dev = text.device
    gap_mean = mean_run * (1 - frac) / frac
    k = int(n / (mean_run + gap_mean) * 1.1) + 16
    gaps = _geometric(g, k, gap_mean, dev)
    runs = _geometric(g, k, mean_run, dev)
    ends = torch.cumsum(gaps + runs, 0)
    starts = ends - runs
    bounds = torch.stack([starts, ends], 1).reshape(-1)        # sorted: s0 < e0 <= s1 < e1 ...
    for lo in range(0, n, _CHUNK):
        m = min(_CHUNK, n - lo)
        pos = torch.arange(lo, lo + m, device=dev)
        inside = (torch.searchsorted(bounds, pos, right=True) & 1).to(torch.uint8)
        text[lo:lo + m] |= inside * 0x20


def _n_runs(g, text: torch.Tensor, frac: float, count: int, lo_len: float, hi_len: float):
    n = text.numel()
    if count <= 0 or frac <= 0:
        return
    u = torch.rand(count, generator=g, device=text.device, dtype=torch.float64)
    lens = torch.exp(u * (math.log(hi_len) - math.log(lo_len)) + math.log(lo_len))
    lens = (lens * (frac * n / float(lens.sum()))).clamp_(1, n // 8 + 1).to(torch.int64).tolist()
    at = (torch.rand(count, generator=g, device=text.device, dtype=torch.float64)).tolist()
    for ln, a in zip(lens, at):
        s = int(a * (n - ln))
        text[s:s + ln] = 78  # 'N'


def _join(text: torch.Tensor, n_seqs: int):
    """Cut into n_seqs sequences: overwrite n_seqs-1 interior bytes with '%' and append '$'."""
    n = text.numel()
    out = torch.empty(n + 1, dtype=torch.uint8, device=text.device)
    out[:n] = text
    out[n] = 36  # '$'
    starts = [0]
    for i in range(1, n_seqs):
        c = (n * i) // n_seqs
        if 0 < c < n:
            out[c] = 37  # '%'
            starts.append(c + 1)
    return out, starts


def syn_ecoli(n_bases: int = 4_641_652, seed: int = 1, device="cpu"):
    """C2 stand-in: one sequence of iid uniform ACGT."""
    g = _gen(device, seed)
    return _join(_bases(g, n_bases, 0.5, device), 1)


def syn_elegans(n_bases: int = 100_286_401, seed: int = 2, n_seqs: int = 7, device="cpu"):
    """C3 stand-in: A/T 32.25 % each; 50 families of 1-6 kb, 40 copies each at 1 % divergence
    (2000 copies in all, scaled with n_bases); 35 % lowercase in runs of mean 300."""
    g = _gen(device, seed)
    text = _bases(g, n_bases, 0.645, device)
    scale = n_bases / 100_286_401
    for f in range(50):
        L = 1000 + (f * 5000) // 49
        if L * 2 > n_bases:
            continue
        fam = _bases(g, L, 0.645, device)
        _plant_family(g, text, fam, max(2, int(round(40 * scale))), 0.01, 0.01, lowercase=False)
    _mask_runs(g, text, 0.35, 300)
    return _join(text, n_seqs)


def syn_human(n_bases: int = 3_100_000_000, seed: int = 4, n_seqs: int = 24, device="cpu"):
    """C4 stand-in, every count scaled by n_bases / 3.1e9: a 300-bp family (1e6 copies, 10-15 %
    divergence) and a 6-kb family (5e5 mostly truncated copies, 5 %), both soft-masked as a repeat
    masker would leave them; 200 segmental duplications of 10-300 kb at 1 % (not masked); 2e4
    tandem repeats (unit 1-6, 50 b - 10 kb); further lowercase runs (mean 350) to ~50 % masked;
    5 % N in 800 runs with log-uniform lengths 1e2 - 1e7."""
    g = _gen(device, seed)
    text = _bases(g, n_bases, 0.59, device)
    scale = n_bases / 3.1e9
    f300 = _bases(g, 300, 0.55, device)
    _plant_family(g, text, f300, int(1_000_000 * scale), 0.10, 0.15, lowercase=True)
    f6k = _bases(g, 6000, 0.58, device)
    for frag, share in ((300, 0.45), (900, 0.30), (2500, 0.17), (6000, 0.08)):
        off = int(torch.randint(0, 6000 - frag + 1, (1,), generator=g, device=device).item())
        _plant_family(g, text, f6k[off:off + frag], int(500_000 * scale * share), 0.05, 0.05, lowercase=True)
    n_dup = max(1, int(round(200 * scale)))
    for _ in range(n_dup):
        r = torch.rand(3, generator=g, device=device, dtype=torch.float64).tolist()
        ln = min(int(10_000 + r[0] * 290_000), n_bases // 8)
        if ln < 64:
            continue
        src = int(r[1] * (n_bases - ln)); dst = int(r[2] * (n_bases - ln))
        seg = (text[src:src + ln] & 0xDF).clone()
        hit = torch.rand(ln, generator=g, device=device) < 0.01
        rnd = _lut(device)[torch.randint(0, 4, (ln,), generator=g, device=device)]
        text[dst:dst + ln] = torch.where(hit, rnd, seg)
    for length, share in ((50, 0.5), (200, 0.3), (1000, 0.15), (10_000, 0.05)):
        _plant_tandem(g, text, max(1, int(20_000 * scale * share)), length, 0.7)
    _mask_runs(g, text, 0.30, 350)
    _n_runs(g, text, 0.05, max(3, int(800 * scale)), 100.0, max(1000.0, 1e7 * scale))
    return _join(text, n_seqs)


def adversarial(kind: str, n: int = 4096, seed: int = 0) -> np.ndarray:
    """Micro-inputs of SURVEY.md 8d: all-A, (ACGT)^k, Fibonacci string, identical sequences, N run,
    tandem arrays.  numpy, always '$'-terminated."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rb = lambda m: acgt[rng.integers(0, 4, size=m)]
    if kind == "all_a":
        body = np.full(n, ord("A"), dtype=np.uint8)
    elif kind == "acgt_k":
        body = np.resize(acgt, n)
    elif kind == "fib":
        a, b = b"A", b"AC"
        while len(b) < n:
            a, b = b, b + a
        body = np.frombuffer(b[:n], dtype=np.uint8).copy()
    elif kind == "two_identical":
        half = rb(n // 2)
        body = np.concatenate([half, np.frombuffer(b"%", dtype=np.uint8), half])
    elif kind == "n_run":
        body = rb(n)
        body[n // 3:n // 3 + max(1000, n // 4)] = ord("N")
    elif kind == "tandem":
        body = rb(n)
        for u in range(1, 7):
            a = (u * n) // 8
            body[a:a + n // 10] = np.resize(rb(u), n // 10)
    else:
        raise ValueError(kind)
    return np.concatenate([body, np.frombuffer(b"$", dtype=np.uint8)])
