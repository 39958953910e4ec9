"""Size-independent checks of an SA/LCP pair on the GPU (tests and bench.py; torch only: nothing else is built).

What a build of the reference guarantees (sufr_builder.rs:346-382, 446-449, 634-767, 893-902) and what can be
checked without a second build at any size:

  * SA is a permutation of the eligible positions        -> count, sum and a weighted checksum (mod 2^64)
  * suffix SA[r-1] < suffix SA[r]                        -> the first differing character, end of text lowest
  * LCP[r] = exact common prefix of SA[r-1], SA[r]       -> compared character by character, UNBOUNDED: pairs that
    stay equal are walked in growing chunks until they differ (megabase `N` runs included)

on sampled adjacent ranks, stratified so that the ranks the re-keying levels produce (LCP beyond the packed key)
are sampled on their own.
"""
from __future__ import annotations

import torch


def normalize_lut(device, ignore_softmask: bool) -> torch.Tensor:
    """reference text map (sufr_builder.rs:144-160) as a 256-entry table"""
    lut = torch.arange(256, dtype=torch.uint8, device=device)
    low = torch.arange(97, 123, device=device)
    lut[low] = 78 if ignore_softmask else (low & 0x5F).to(torch.uint8)
    return lut


def eligible_table(device, is_dna: bool, allow_ambiguity: bool) -> torch.Tensor:
    """suffix-start predicate on NORMALISED bytes (sufr_builder.rs:446-449)"""
    el = torch.ones(256, dtype=torch.bool, device=device)
    if is_dna and not allow_ambiguity:
        el[:] = False
        for c in b"ACGT$":
            el[c] = True
    return el


def check_permutation(raw: torch.Tensor, sa: torch.Tensor, *, is_dna: bool, allow_ambiguity: bool = False,
                      ignore_softmask: bool = False, raw_is_normalised: bool = False) -> int:
    """SA holds every eligible position exactly once (count + two checksums).  Returns the count."""
    dev = raw.device
    n = raw.numel()
    lut = normalize_lut(dev, ignore_softmask)
    el = eligible_table(dev, is_dna, allow_ambiguity)
    cnt = 0; tot = 0; wtot = 0; xtot = 0
    for lo in range(0, n, 1 << 28):
        blk = raw[lo:lo + (1 << 28)]
        if not raw_is_normalised:
            blk = lut[blk.long()]
        m = el[blk.long()]
        pos = torch.arange(lo, lo + blk.numel(), device=dev)[m]
        cnt += int(m.sum()); tot += int(pos.sum()); wtot += int((pos * (pos % 1009)).sum())
        xtot ^= int(_xor_reduce(pos * 0x9E3779B1 + 12345))
    M64 = (1 << 64) - 1
    assert sa.numel() == cnt, f"SA holds {sa.numel()} entries, the text has {cnt} suffix starts"
    stot = 0; swtot = 0; sxtot = 0
    for lo in range(0, sa.numel(), 1 << 28):               # all sums modulo 2^64, block by block
        p64 = _values(sa[lo:lo + (1 << 28)])
        assert int(p64.min()) >= 0 and int(p64.max()) < n, "SA entry outside the text"
        stot += int(p64.sum()); swtot += int((p64 * (p64 % 1009)).sum())
        sxtot ^= int(_xor_reduce(p64 * 0x9E3779B1 + 12345))
    assert (stot & M64) == (tot & M64), "SA is not a permutation of the eligible positions (sum)"
    assert (swtot & M64) == (wtot & M64), "SA is not a permutation (weighted sum)"
    assert sxtot == xtot, "SA is not a permutation (xor of hashes)"
    return cnt


def _values(v: torch.Tensor) -> torch.Tensor:
    """u32 values stored in int32 tensors, or int64 tensors as they are (index_width 8)"""
    return v if v.dtype == torch.int64 else v.to(torch.int64) & 0xFFFFFFFF


def _xor_reduce(v: torch.Tensor) -> torch.Tensor:
    v = v.reshape(-1)
    while v.numel() > 1:
        if v.numel() % 2:
            v = torch.cat([v, v.new_zeros(1)])
        v = v[0::2] ^ v[1::2]
    return v[0] if v.numel() else v.new_zeros(())


def exact_lcp_pairs(norm: torch.Tensor, a: torch.Tensor, b: torch.Tensor):
    """Exact common prefix of suffixes a[i], b[i] of the normalised text and whether suffix a[i] sorts before
    b[i] (a suffix that is a proper prefix of the other sorts first).  Unbounded: still-equal pairs are walked in
    chunks that grow as the active set shrinks."""
    dev = norm.device
    n = norm.numel()
    m = a.numel()
    lcp = torch.zeros(m, dtype=torch.int64, device=dev)
    less = torch.zeros(m, dtype=torch.bool, device=dev)
    active = torch.arange(m, device=dev)
    off = torch.zeros(m, dtype=torch.int64, device=dev)
    budget = 1 << 26                                     # characters gathered per step and side
    while active.numel():
        W = int(min(max(64, budget // active.numel()), 1 << 22))
        ar = torch.arange(W, device=dev)
        ia = a[active, None] + off[active, None] + ar[None, :]
        ib = b[active, None] + off[active, None] + ar[None, :]
        ta = torch.where(ia < n, norm[ia.clamp(max=n - 1)].to(torch.int16), torch.full((), -1, dtype=torch.int16, device=dev))
        tb = torch.where(ib < n, norm[ib.clamp(max=n - 1)].to(torch.int16), torch.full((), -1, dtype=torch.int16, device=dev))
        diff = ta != tb
        anyd = diff.any(1)
        first = diff.to(torch.uint8).argmax(1)
        done = active[anyd]
        lcp[done] = off[done] + first[anyd]
        ca = ta[anyd].gather(1, first[anyd, None])[:, 0]
        cb = tb[anyd].gather(1, first[anyd, None])[:, 0]
        less[done] = ca < cb                              # -1 (past the end) is below every character
        rest = active[~anyd]
        off[rest] += W
        # both suffixes past the end of the text without a difference cannot happen for a != b
        active = rest
    return lcp, less


def check_sampled_ranks(norm: torch.Tensor, sa: torch.Tensor, lcp: torch.Tensor, *, samples: int = 1_000_000,
                        deep_samples: int = 100_000, deep_min_lcp: int = 64, seed: int = 5) -> dict:
    """Order and exact (unbounded) LCP on `samples` adjacent ranks, `deep_samples` of them drawn from the ranks
    whose LCP is at least `deep_min_lcp` (the ones the levels beyond the packed key produce)."""
    dev = norm.device
    s = sa.numel()
    out = {"ranks": 0, "deep_ranks": 0, "max_lcp_checked": 0}
    if s < 2:
        return out
    g = torch.Generator(device=dev); g.manual_seed(seed)
    picks = [torch.randint(1, s, (min(samples, 4 * s),), generator=g, device=dev)]
    ndeep = 0
    if deep_samples:
        deep = []
        for lo in range(0, s, 1 << 28):                  # ranks with a long LCP, block by block
            d = (_values(lcp[lo:lo + (1 << 28)]) >= deep_min_lcp).nonzero()[:, 0] + lo
            deep.append(d[d > 0])
        deep = torch.cat(deep)
        if deep.numel():
            sel = torch.randint(0, deep.numel(), (min(deep_samples, deep.numel()),), generator=g, device=dev)
            picks.append(deep[sel]); ndeep = int(sel.numel())
    pick = torch.cat(picks)
    a = _values(sa[pick - 1])
    b = _values(sa[pick])
    want = _values(lcp[pick])
    for lo in range(0, pick.numel(), 1 << 18):
        got, less = exact_lcp_pairs(norm, a[lo:lo + (1 << 18)], b[lo:lo + (1 << 18)])
        w = want[lo:lo + (1 << 18)]
        bad = (got != w) | ~less
        if bool(bad.any()):
            i = int(bad.nonzero()[0, 0])
            raise AssertionError(f"rank {int(pick[lo + i])}: suffixes {int(a[lo + i])}, {int(b[lo + i])} "
                                 f"LCP {int(w[i])} in the array, {int(got[i])} in the text, in order: {bool(less[i])}")
        out["max_lcp_checked"] = max(out["max_lcp_checked"], int(got.max()))
    assert int(lcp[0]) == 0, "LCP[0] must be 0"
    out["ranks"] = int(pick.numel()); out["deep_ranks"] = ndeep
    return out
