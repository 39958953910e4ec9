"""Windowed builds (sufr_wide.inc): texts the 32-bit records cannot address are built in
overlapping windows and merged by rank on the device.  With a forced small window the same code runs on the reference's
own inputs, where the result must be the golden file byte for byte and the oracle's arrays bit for bit; at full size
(more than 2^32 bytes) the answer is checked through properties: permutation of the suffix starts, order and exact LCP
on sampled ranks."""
import subprocess

import numpy as np
import pytest
import torch

import sufr_amd
from sufr_amd import synth
import gpu_verify as verify
from oracle_helper import GOLDEN, GOLDEN_CASES

pytestmark = pytest.mark.gpu


def build(x, window, margin, index_width, **flags):
    db = sufr_amd.DeviceBuilder(0)
    db.ctx.set_window(window, margin)
    sa, lcp = db.sort(x, index_width=index_width, **flags)
    sa, lcp = sa.cpu().numpy(), lcp.cpu().numpy()
    db.close()
    if index_width == 4:
        sa, lcp = sa.view(np.uint32), lcp.view(np.uint32)
    return sa.astype(np.uint64), lcp.astype(np.uint64), db.stats


def repeat_text(n, seed, repeat_len, copies):
    rng = np.random.default_rng(seed)
    t = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
    seg = t[100:100 + repeat_len].copy()
    for k in range(copies):
        at = int(rng.integers(0, n - repeat_len - 1))
        t[at:at + repeat_len] = seg
    t[n // 3: n // 3 + 700] = ord("N")
    t[-1] = ord("$")
    return t


@pytest.mark.parametrize("index_width", [4, 8])
@pytest.mark.parametrize("window,margin", [(5000, 3000), (4096, 64), (20000, 100), (4000, 16)])
@pytest.mark.parametrize("flags", [dict(is_dna=True), dict(is_dna=True, allow_ambiguity=True), dict()])
def test_windowed_build_equals_oracle(oracle, window, margin, index_width, flags):
    """60 kb with 40 copies of a 900-symbol repeat and an N run: repeats cross every window boundary, and margins shorter
    than the repeat force the retry with the widest margin"""
    t = repeat_text(60_000, 1, 900, 40)
    want_sa, want_lcp, _ = oracle.build(t, **flags)
    sa, lcp, st = build(torch.from_numpy(t).cuda(), window, margin, index_width, **flags)
    assert np.array_equal(sa, want_sa.astype(np.uint64))
    assert np.array_equal(lcp, want_lcp.astype(np.uint64))
    assert st.num_suffixes == sa.size and st.text_len == t.size


@pytest.mark.parametrize("index_width", [4, 8])
def test_two_windows_with_a_one_sided_repeat(oracle, index_width):
    """Two windows where thousands of suffixes of one window fall between two neighbouring suffixes of the other (a
    homopolymer and a tandem array that live in the second window only): the ranges above the sequential-merge limit
    take the binary-search path"""
    rng = np.random.default_rng(6)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    t = np.concatenate([acgt[rng.integers(0, 4, 30_000)], np.full(12_000, ord("A"), dtype=np.uint8), acgt[rng.integers(0, 4, 3_000)],
                        np.resize(np.frombuffer(b"ACG", dtype=np.uint8), 9_000), acgt[rng.integers(0, 4, 6_000)],
                        np.frombuffer(b"$", dtype=np.uint8)])
    want_sa, want_lcp, _ = oracle.build(t, is_dna=True)
    sa, lcp, _ = build(torch.from_numpy(t).cuda(), 30_000, 4_000, index_width, is_dna=True)
    assert np.array_equal(sa, want_sa.astype(np.uint64)) and np.array_equal(lcp, want_lcp.astype(np.uint64))


def test_windowed_build_of_raw_soft_masked_text(oracle):
    rng = np.random.default_rng(2)
    t = repeat_text(50_000, 3, 300, 30)
    low = rng.random(t.size) < 0.3
    t[low & (t != ord("$"))] |= 0x20
    for soft in (False, True):
        norm = oracle.normalize(t, soft)
        want_sa, want_lcp, _ = oracle.build(norm, is_dna=True)
        sa, lcp, _ = build(torch.from_numpy(t).cuda(), 7000, 500, 8, is_dna=True, ignore_softmask=soft, raw_text=True)
        assert np.array_equal(sa, want_sa.astype(np.uint64)) and np.array_equal(lcp, want_lcp.astype(np.uint64))


@pytest.mark.parametrize("name", ["long_dna_sequence.sufr", "long_dna_sequence_allow_ambiguity.sufr", "uniprot.sufr", "3.sufr", "2.sufr"])
def test_cli_create_in_windows_writes_the_golden_file(tmp_path, name):
    """`sufr create --window` on the reference's inputs: the file is the reference's golden file (mk_test_files.py:63-89)"""
    case = GOLDEN_CASES[name]
    out = tmp_path / name
    n = (GOLDEN / "expected" / name).stat().st_size
    args = [str(sufr_amd.CLI_PATH), "create", str(GOLDEN / "inputs" / case["fa"]), "-o", str(out), "-n", "16", "-r", "42",
            "--window", str(max(16, n // 40)), "--margin", "48"]
    if case.get("is_dna"):
        args.append("-d")
    if case.get("allow_ambiguity"):
        args.append("-a")
    if case.get("ignore_softmask"):
        args.append("-i")
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == (GOLDEN / "expected" / name).read_bytes()


@pytest.mark.parametrize("index_width", [4, 8])
@pytest.mark.parametrize("window,margin", [(4096, 64), (5000, 3000), (20000, 100), (4000, 8)])
def test_seed_mask_build_in_windows_equals_oracle(oracle, window, margin, index_width):
    """--seed-mask across windows (find_lcp's mask arm, sufr_builder.rs:272-300): the windows are built with the mask and
    merged under the masked order; margins shorter than the mask are widened to its span"""
    t = repeat_text(60_000, 1, 900, 40)
    for mask in ["1101", "10111011", "111010010100110111"]:
        want_sa, want_lcp, _ = oracle.build(t, is_dna=True, seed_mask=mask, threads=8)
        sa, lcp, st = build(torch.from_numpy(t).cuda(), window, margin, index_width, is_dna=True, seed_mask=mask)
        assert np.array_equal(sa, want_sa.astype(np.uint64)), mask
        assert np.array_equal(lcp, want_lcp.astype(np.uint64)), mask
        assert st.num_suffixes == sa.size


@pytest.mark.parametrize("index_width", [4, 8])
@pytest.mark.parametrize("window,margin", [(4096, 64), (5000, 3000), (20000, 100), (30000, 16)])
@pytest.mark.parametrize("L", [1, 8, 21, 100, 1500])
def test_max_query_len_build_in_windows_is_the_canonical_form(window, margin, index_width, L):
    """--max-query-len across windows: the same canonical member of the reference's family as the one-window build
    (tests/test_gpu_parity.py::test_max_query_len_canonical_form) -- first L symbols in order, ties in descending position
    over the WHOLE text, LCP capped at L.  L above the margin exercises the retry with the widest margin; L = 1500 is
    longer than the 900-symbol repeat, i.e. the plain order"""
    t = repeat_text(60_000, 1, 900, 40)
    x = torch.from_numpy(t).cuda()
    one_sa, one_lcp, _ = build(x, 0, 0, index_width, is_dna=True, max_query_len=L)
    sa, lcp, _ = build(x, window, margin, index_width, is_dna=True, max_query_len=L)
    assert np.array_equal(sa, one_sa) and np.array_equal(lcp, one_lcp)
    full_sa, full_lcp, _ = build(x, 0, 0, index_width, is_dna=True)
    assert np.array_equal(lcp, np.minimum(full_lcp, L))
    tie = lcp >= L
    assert np.all(sa[1:][tie[1:]] < sa[:-1][tie[1:]])


def test_masked_and_truncated_builds_of_a_protein_text_in_three_windows(oracle):
    rng = np.random.default_rng(9)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    t = aa[rng.integers(0, 20, 30_000)].copy()
    t[5000:9000] = t[20000:24000]
    t[::997] = ord("%")
    t[-1] = ord("$")
    want_sa, want_lcp, _ = oracle.build(t, seed_mask="110101", threads=8)
    sa, lcp, _ = build(torch.from_numpy(t).cuda(), 11_000, 300, 8, seed_mask="110101")
    assert np.array_equal(sa, want_sa.astype(np.uint64)) and np.array_equal(lcp, want_lcp.astype(np.uint64))
    one_sa, one_lcp, _ = build(torch.from_numpy(t).cuda(), 0, 0, 8, max_query_len=5)
    sa, lcp, _ = build(torch.from_numpy(t).cuda(), 11_000, 300, 8, max_query_len=5)
    assert np.array_equal(sa, one_sa) and np.array_equal(lcp, one_lcp)


def test_medium_windowed_build_properties():
    """40 Mb stand-in genome in 7 windows: permutation + order + exact LCP on 200 000 ranks, and equality with the
    single-window build of the same text"""
    x, _ = synth.syn_human(40_000_000, seed=4, device="cuda")
    db = sufr_amd.DeviceBuilder(0)
    want_sa, want_lcp = db.sort(x, is_dna=True, raw_text=True)
    db.ctx.set_window(6_000_000, 1 << 20)
    sa, lcp = db.sort(x, is_dna=True, raw_text=True, index_width=8)
    assert torch.equal(sa, want_sa.to(torch.int64) & 0xFFFFFFFF)
    assert torch.equal(lcp, want_lcp.to(torch.int64) & 0xFFFFFFFF)
    db.close()


def test_run_buckets_inside_windows_equal_the_single_build_and_the_oracle(oracle):
    """--allow-ambiguity text whose N bucket is ordered in closed form (sufr_runs.inc: 2-3 M records per window), built in two
    windows: N runs cross the cut and sit in the margin, whose suffixes are dropped afterwards; merged arrays = the one-window
    arrays = the oracle's (runs stay below 1 000 symbols so that the byte-walking reference is the checker)"""
    rng = np.random.default_rng(23)
    n = 44_000_000
    t = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
    at = 0
    while at < n - 3000:
        at += int(rng.integers(200, 6000)); ln = int(rng.integers(21, 950))
        t[at:at + ln] = ord("N"); at += ln
    t[n // 2 - 300:n // 2 + 400] = ord("N")                     # one run across the cut
    t[-1] = ord("$")
    x = torch.from_numpy(t).cuda()
    flags = dict(is_dna=True, allow_ambiguity=True)
    db = sufr_amd.DeviceBuilder(0)
    one_sa, one_lcp = db.sort(x, **flags)
    one_sa = one_sa.cpu().numpy().view(np.uint32).astype(np.uint64); one_lcp = one_lcp.cpu().numpy().view(np.uint32).astype(np.uint64)
    db.close()
    sa, lcp, st = build(x, n // 2, 1 << 20, 8, **flags)
    assert np.array_equal(sa, one_sa) and np.array_equal(lcp, one_lcp)
    import os
    want_sa, want_lcp, _ = oracle.build(t, threads=min(32, os.cpu_count() or 1), **flags)
    assert np.array_equal(one_sa, want_sa.astype(np.uint64)) and np.array_equal(one_lcp, want_lcp.astype(np.uint64))


def test_text_beyond_32_bits():
    """4.4e9 bytes (> 2^32): random DNA with repeats planted across the window boundary, u64 arrays.  Checked without a
    second build: SA is a permutation of the suffix starts, sampled neighbours are in order with the exact LCP."""
    import gc
    gc.collect()
    torch.cuda.empty_cache()                                  # ~200 GB of HBM: nothing of earlier tests may linger in torch's cache
    n = 4_400_000_001
    dev = "cuda"
    g = torch.Generator(device=dev); g.manual_seed(8)
    x = torch.empty(n, dtype=torch.uint8, device=dev)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    for lo in range(0, n, 1 << 28):
        m = min(1 << 28, n - lo)
        x[lo:lo + m] = lut[torch.randint(0, 4, (m,), generator=g, device=dev)]
    half = n // 2
    seg = x[1000:1000 + 50_000].clone()
    for at in (half - 20_000, half + 3_000_000, n - 60_000, 2_000_000_000, 4_300_000_000):   # repeats on both sides of 2^32 and of the cut
        x[at:at + seg.numel()] = seg
    x[half - 5_000_000:half - 4_999_000] = ord("N")
    x[-1] = ord("$")
    db = sufr_amd.DeviceBuilder(0)
    with pytest.raises(sufr_amd.SufrHipError):                # the u32 entry point cannot hold these positions
        db.sort(x[:4_300_000_000], is_dna=True)
    try:
        sa, lcp = db.sort(x, is_dna=True, index_width=8)
    except (sufr_amd.SufrHipError, torch.OutOfMemoryError) as e:      # ~200 GB of HBM: a box that cannot spare it skips, loudly
        if isinstance(e, sufr_amd.SufrHipError) and e.code != -4:
            raise
        pytest.skip(f"not enough free HBM for the 4.4e9-byte build: {e}")
    st = db.stats
    assert sa.dtype == torch.int64 and st.text_len == n
    count = verify.check_permutation(x, sa, is_dna=True, raw_is_normalised=True)
    assert count == sa.numel() == st.num_suffixes
    assert int(sa.max()) > (1 << 32)
    res = verify.check_sampled_ranks(x, sa, lcp, samples=400_000, deep_samples=100_000, deep_min_lcp=40)
    assert res["deep_ranks"] > 0 and res["max_lcp_checked"] >= 40_000
    print(f"4.4e9-byte text: {st.ms_total:.0f} ms device total, {res}")
    # round 5: the same build as TWO shards (what two ranks run): each shard equals its slice of the single build, element for
    # element, apart from the first LCP of shard 1, which the device stitch sets from the boundary triples
    from sufr_amd import shards
    host_sa, host_lcp = sa.cpu(), lcp.cpu()
    del sa, lcp
    torch.cuda.empty_cache()
    off, rows, firsts = 0, [], []
    out_sa = torch.empty(n // 2 + (1 << 28), dtype=torch.int64, device=dev)
    out_lcp = torch.empty_like(out_sa)
    for r in range(2):
        psa, plcp = db.sort(x, is_dna=True, index_width=8, shard_index=r, num_shards=2, out_sa=out_sa, out_lcp=out_lcp)
        k = psa.numel()
        assert k > 0 and torch.equal(psa.cpu(), host_sa[off:off + k]) and torch.equal(plcp[1:].cpu(), host_lcp[off + 1:off + k])
        rows.append(shards.gather_boundaries_device(psa, k))
        if r == 1:
            bounds = torch.cat(rows).contiguous()
            shards.stitch_device(db.ctx, n, bounds, 1, plcp)
            db.ctx.synchronize()
            assert int(plcp[0]) == int(host_lcp[off])
        off += k
    assert off == count
    print(f"4.4e9-byte text as two shards: {db.stats.ms_total:.0f} ms device total for the second shard")
    del out_sa, out_lcp, psa, plcp, host_sa, host_lcp
    # the same text under the reference's `hu-mask` seed (Makefile:82): windows built with the mask, merged under the masked
    # order.  Sampled neighbours: care symbols in order, equal ones in descending position, LCP = equal care symbols.
    torch.cuda.empty_cache()
    mask = "111010010100110111"
    try:
        msa, mlcp = db.sort(x, is_dna=True, index_width=8, seed_mask=mask)
    except (sufr_amd.SufrHipError, torch.OutOfMemoryError) as e:
        if isinstance(e, sufr_amd.SufrHipError) and e.code != -4:
            raise
        pytest.skip(f"not enough free HBM for the masked 4.4e9-byte build: {e}")
    assert verify.check_permutation(x, msa, is_dna=True, raw_is_normalised=True) == msa.numel() == count
    offs = torch.tensor([i for i, c in enumerate(mask) if c == "1"], device=dev)
    g2 = torch.Generator(device=dev); g2.manual_seed(11)
    pick = torch.randint(1, msa.numel(), (400_000,), generator=g2, device=dev)
    a, b = msa[pick - 1], msa[pick]
    ok = (a + len(mask) < n) & (b + len(mask) < n)            # (the last few suffixes: care symbols past the end, covered at 60 kb)
    a, b, w = a[ok], b[ok], mlcp[pick][ok]
    ka, kb = x[a[:, None] + offs[None, :]].to(torch.int16), x[b[:, None] + offs[None, :]].to(torch.int16)
    diff = ka != kb
    anyd = diff.any(1)
    first = torch.where(anyd, diff.to(torch.uint8).argmax(1), torch.full_like(a, offs.numel()))
    assert bool((w == first).all())
    fd = first.clamp(max=offs.numel() - 1)[:, None]
    in_order = torch.where(anyd, ka.gather(1, fd)[:, 0] < kb.gather(1, fd)[:, 0], b < a)
    assert bool(in_order.all())
    assert int((~anyd).sum()) > 1000                          # ties across the whole text were among the samples
    print(f"4.4e9-byte text, seed mask: {db.stats.ms_total:.0f} ms device total, {int((~anyd).sum())} of {int(ok.sum())} sampled pairs tie")
    db.close()
    del x, msa, mlcp
    torch.cuda.empty_cache()


def test_cli_creates_masked_and_truncated_files_in_windows(tmp_path):
    """`sufr create --window ... -m / -s`: the file of the windowed build is the file of the one-window build, byte for byte
    (for the mask that is the reference's own family member: uniprot-masked.sufr is pinned in test_gpu_parity.py)"""
    fa = GOLDEN / "inputs" / "long_dna_sequence.fa"
    for tag, extra in (("m", ["-m", "8"]), ("s", ["-s", "1101"])):
        one, win = tmp_path / f"one_{tag}.sufr", tmp_path / f"win_{tag}.sufr"
        r = subprocess.run([str(sufr_amd.CLI_PATH), "create", str(fa), "-d", "-o", str(one), *extra], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        r = subprocess.run([str(sufr_amd.CLI_PATH), "create", str(fa), "-d", "-o", str(win), "--window", "4096", "--margin", "48", *extra],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert win.read_bytes() == one.read_bytes()
    r = subprocess.run([str(sufr_amd.CLI_PATH), "create", str(fa), "-d", "-o", str(tmp_path / "y.sufr"), "--window", "4096", "-m", "0"],
                       capture_output=True, text=True)                                   # Some(0): a plain build
    assert r.returncode == 0, r.stderr

# ---- round 5: a repeat longer than the WIDEST margin -------------------------------------------------
def build_capped(x, window, margin, retry, index_width, **flags):
    """like build(), with the re-build margin capped (sufr_hip_set_window_retry): repeats longer than `retry` take the
    whole-text repair (sufr_wide.inc, repair_window); returns the number of suffixes it ordered too"""
    db = sufr_amd.DeviceBuilder(0)
    db.ctx.set_window(window, margin)
    db.ctx.set_window_retry(retry)
    sa, lcp = db.sort(x, index_width=index_width, **flags)
    sa, lcp = sa.cpu().numpy(), lcp.cpu().numpy()
    repaired = db.ctx.window_repairs
    db.close()
    if index_width == 4:
        sa, lcp = sa.view(np.uint32), lcp.view(np.uint32)
    return sa.astype(np.uint64), lcp.astype(np.uint64), repaired


@pytest.mark.parametrize("index_width", [4, 8])
@pytest.mark.parametrize("window,margin,retry", [(5000, 300, 100), (4096, 64, 64), (8000, 100, 500), (20000, 16, 16)])
@pytest.mark.parametrize("flags", [dict(is_dna=True), dict(is_dna=True, allow_ambiguity=True), dict()])
def test_repeat_longer_than_the_widest_margin_is_ordered_over_the_whole_text(oracle, window, margin, retry, index_width, flags):
    """40 copies of a 900-symbol repeat cross every window boundary and the re-build margin is capped BELOW the repeat: rounds
    2-4 returned SUFR_HIP_E_UNSUPPORTED here ("true 64-bit records are not built yet").  Now the suffixes the window saw only
    a prefix of are ordered by whole-text comparison with 64-bit positions and merged back: the oracle's arrays, bit for bit."""
    t = repeat_text(60_000, 1, 900, 40)
    want_sa, want_lcp, _ = oracle.build(t, **flags)
    sa, lcp, repaired = build_capped(torch.from_numpy(t).cuda(), window, margin, retry, index_width, **flags)
    assert repaired > 0 or retry > 100, "the cap did not force the repair path"     # (a 500-symbol cap: only if a copy lies just so)
    assert np.array_equal(sa, want_sa.astype(np.uint64))
    assert np.array_equal(lcp, want_lcp.astype(np.uint64))


def test_long_exact_duplicates_across_capped_windows(oracle):
    """exact copies of a 3 000-symbol segment 3 500 symbols apart (both inside one window + margin, the second one cut by its
    end) and a 3 000-symbol homopolymer that ends behind a capped margin, windows of ~6 250 with re-build margins of at most
    400: thousands of suffixes tie through the end of their window, with common prefixes of thousands of symbols"""
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    t = acgt[rng.integers(0, 4, 50_000)].copy()
    seg = t[1_000:4_000].copy()
    for at in (4_500, 30_100):
        t[at:at + seg.size] = seg
    t[41_000:44_000] = ord("T")
    t[-1] = ord("$")
    want_sa, want_lcp, _ = oracle.build(t, is_dna=True)
    for index_width in (4, 8):
        sa, lcp, repaired = build_capped(torch.from_numpy(t).cuda(), 7_000, 200, 400, index_width, is_dna=True)
        assert repaired > 1_000
        assert np.array_equal(sa, want_sa.astype(np.uint64)) and np.array_equal(lcp, want_lcp.astype(np.uint64))


@pytest.mark.parametrize("L", [50, 700, 2_000])
def test_max_query_len_above_the_capped_margin_equals_the_one_window_build(L):
    """-m L with L above the capped re-build margin: the suffixes with fewer than L symbols inside their window are ordered
    over the whole text under the capped order (first L symbols, then descending position)"""
    t = repeat_text(60_000, 5, 900, 40)
    x = torch.from_numpy(t).cuda()
    db = sufr_amd.DeviceBuilder(0)
    one_sa, one_lcp = (a.cpu().numpy().view(np.uint32).astype(np.uint64) for a in db.sort(x, is_dna=True, max_query_len=L))
    db.close()
    sa, lcp, repaired = build_capped(x, 6_000, 40, 40, 4, is_dna=True, max_query_len=L)
    assert repaired > 0
    assert np.array_equal(sa, one_sa) and np.array_equal(lcp, one_lcp)


def test_repairs_are_zero_when_the_margin_suffices(oracle):
    t = repeat_text(60_000, 1, 900, 40)
    sa, lcp, repaired = build_capped(torch.from_numpy(t).cuda(), 5000, 3000, 0, 4, is_dna=True)
    want_sa, want_lcp, _ = oracle.build(t, is_dna=True)
    assert repaired == 0 and np.array_equal(sa, want_sa.astype(np.uint64)) and np.array_equal(lcp, want_lcp.astype(np.uint64))


# ---- round 5: shards of a windowed build --------------------------------------------------------------
def sharded_windowed(x, n, num_shards, window, margin, index_width, retry=0, **flags):
    """every shard of a windowed build on one context, one after the other (what the ranks of an N-GPU job run), stitched on
    the device with the boundary triples in device memory; returns the concatenated arrays, the shard sizes and the repairs"""
    from sufr_amd import shards
    db = sufr_amd.DeviceBuilder(0)
    db.ctx.set_window(window, margin)
    db.ctx.set_window_retry(retry)
    parts, rows, repaired = [], [], 0
    for r in range(num_shards):
        sa, lcp = db.sort(x, index_width=index_width, shard_index=r, num_shards=num_shards, **flags)
        repaired += db.ctx.window_repairs
        parts.append((sa.clone(), lcp.clone()))
        rows.append(shards.gather_boundaries_device(sa, sa.numel()))
    bounds = torch.cat(rows).contiguous()
    for r in range(num_shards):
        shards.stitch_device(db.ctx, n, bounds, r, parts[r][1])
    db.ctx.synchronize()
    sa = torch.cat([p[0] for p in parts]).cpu().numpy(); lcp = torch.cat([p[1] for p in parts]).cpu().numpy()
    db.close()
    if index_width == 4:
        sa, lcp = sa.view(np.uint32), lcp.view(np.uint32)
    return sa.astype(np.uint64), lcp.astype(np.uint64), [int(p[0].numel()) for p in parts], repaired


@pytest.mark.parametrize("index_width", [4, 8])
@pytest.mark.parametrize("num_shards,window,margin", [(2, 20000, 3000), (3, 7000, 500), (5, 5000, 64), (8, 30000, 100)])
@pytest.mark.parametrize("flags", [dict(is_dna=True), dict(is_dna=True, allow_ambiguity=True), dict()])
def test_sharded_windowed_build_concatenates_to_the_oracle(oracle, num_shards, window, margin, index_width, flags):
    """2 shards x 3 windows (and 3 x 9, 5 x 12, 8 x 2): rounds 2-4 refused a sharded windowed build ("windowed builds are
    single-GPU").  The shards -- ranges of the first 8 bytes, the same on every rank -- concatenate, after the device stitch of
    their first LCPs, to the oracle's arrays bit for bit."""
    t = repeat_text(60_000, 1, 900, 40)
    want_sa, want_lcp, _ = oracle.build(t, **flags)
    sa, lcp, sizes, _ = sharded_windowed(torch.from_numpy(t).cuda(), t.size, num_shards, window, margin, index_width, **flags)
    assert sum(sizes) == want_sa.size and sum(1 for c in sizes if c) >= min(num_shards, 4)      # (DNA: few distinct first bytes)
    assert np.array_equal(sa, want_sa.astype(np.uint64))
    assert np.array_equal(lcp, want_lcp.astype(np.uint64))


def test_sharded_windowed_build_with_repairs_and_length_cap(oracle):
    """the shards of a windowed build whose windows need the whole-text repair (re-build margin capped below the repeat), and
    of a -m 50 build in windows: both concatenate to the one-GPU arrays"""
    t = repeat_text(60_000, 1, 900, 40)
    x = torch.from_numpy(t).cuda()
    want_sa, want_lcp, _ = oracle.build(t, is_dna=True)
    sa, lcp, sizes, repaired = sharded_windowed(x, t.size, 3, 4096, 64, 8, retry=64, is_dna=True)
    assert repaired > 0 and np.array_equal(sa, want_sa.astype(np.uint64)) and np.array_equal(lcp, want_lcp.astype(np.uint64))
    db = sufr_amd.DeviceBuilder(0)
    one_sa, one_lcp = (a.cpu().numpy().view(np.uint32).astype(np.uint64) for a in db.sort(x, is_dna=True, max_query_len=50))
    db.close()
    sa, lcp, sizes, _ = sharded_windowed(x, t.size, 4, 9000, 100, 4, is_dna=True, max_query_len=50)
    assert np.array_equal(sa, one_sa) and np.array_equal(lcp, one_lcp)


def test_sharded_windowed_seed_mask_is_refused_with_a_message():
    t = repeat_text(60_000, 1, 900, 40)
    db = sufr_amd.DeviceBuilder(0)
    db.ctx.set_window(9000, 100)
    with pytest.raises(sufr_amd.SufrHipError) as e:
        db.sort(torch.from_numpy(t).cuda(), is_dna=True, seed_mask="1101", shard_index=1, num_shards=2)
    assert e.value.code == -6 and "one GPU" in e.value.message
    db.close()


def test_sharded_windowed_build_with_empty_shards(oracle):
    """a homopolymer: every sampled 8-byte prefix is the same, so all range bounds coincide and most shards are empty -- they
    build, stitch (nothing to do) and concatenate like the others (profiles/soak_wide.py found the Python stitch passing a null
    pointer for an empty shard)"""
    t = np.full(30_001, ord("A"), dtype=np.uint8)
    t[11_000] = ord("C")
    t[-1] = ord("$")
    want_sa, want_lcp, _ = oracle.build(t, is_dna=True)
    sa, lcp, sizes, _ = sharded_windowed(torch.from_numpy(t).cuda(), t.size, 5, 9_000, 64, 8, is_dna=True)
    assert sizes.count(0) >= 2 and sum(sizes) == want_sa.size
    assert np.array_equal(sa, want_sa.astype(np.uint64)) and np.array_equal(lcp, want_lcp.astype(np.uint64))


@pytest.mark.parametrize("name", ["long_dna_sequence.sufr", "long_dna_sequence_allow_ambiguity.sufr", "uniprot.sufr", "2.sufr"])
def test_cli_create_in_windows_over_several_contexts_writes_the_golden_file(tmp_path, name):
    """`sufr --devices 0,0,0 create --window`: the shards of a windowed build, one per context, stream their slices into the one
    file (round 5: rounds 2-4 built such texts on the first device alone) -- the reference's golden file byte for byte"""
    case = GOLDEN_CASES[name]
    out = tmp_path / name
    n = (GOLDEN / "expected" / name).stat().st_size
    args = [str(sufr_amd.CLI_PATH), "--devices", "0,0,0", "create", str(GOLDEN / "inputs" / case["fa"]), "-o", str(out), "-n", "16", "-r", "42",
            "--window", str(max(16, n // 40)), "--margin", "48"]
    if case.get("is_dna"):
        args.append("-d")
    if case.get("allow_ambiguity"):
        args.append("-a")
    if case.get("ignore_softmask"):
        args.append("-i")
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == (GOLDEN / "expected" / name).read_bytes()
    assert not (tmp_path / (name + ".partial")).exists()


def test_cli_windowed_create_over_contexts_equals_the_one_context_file(tmp_path):
    """a 60 kb text with repeats across every window boundary, plain and -m 40: two and five contexts write the file one context
    writes (the first LCP of every shard stitched on its device under the order of the build)"""
    t = repeat_text(60_000, 9, 900, 40)
    fa = tmp_path / "r.fa"
    fa.write_bytes(b">r\n" + t[:-1].tobytes() + b"\n")
    for extra in ([], ["-m", "40"]):
        files = []
        for devs in ("0", "0,0", "0,0,0,0,0"):
            out = tmp_path / f"o_{devs.count(',')}_{len(extra)}.sufr"
            r = subprocess.run([str(sufr_amd.CLI_PATH), "--devices", devs, "create", "-d", str(fa), "-o", str(out), "--window", "7000",
                                "--margin", "300"] + extra, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            files.append(out.read_bytes())
        assert files[0] == files[1] == files[2]


# ---- out of core: a windowed create whose arrays leave for the file shard after shard (sufr_hip_set_array_budget) ----

@pytest.mark.parametrize("name", ["long_dna_sequence.sufr", "long_dna_sequence_allow_ambiguity.sufr", "uniprot.sufr", "2.sufr", "1.sufr"])
@pytest.mark.parametrize("devs", ["0", "0,0"])
def test_cli_out_of_core_create_writes_the_golden_file(tmp_path, name, devs):
    """`sufr create --window W --array-budget B`: the arrays never exist as a whole -- ceil(2 n width / B) shards are built one
    after another (two contexts: in rounds of two) and each slice is streamed to its place in the file before the next shard is
    built.  The reference's golden files byte for byte, with a budget of a seventh of the arrays."""
    case = GOLDEN_CASES[name]
    out = tmp_path / name
    golden = (GOLDEN / "expected" / name).read_bytes()
    n = len(golden)
    args = [str(sufr_amd.CLI_PATH), "--devices", devs, "create", str(GOLDEN / "inputs" / case["fa"]), "-o", str(out), "-n", "16", "-r", "42",
            "--window", str(max(16, n // 40)), "--margin", "48", "--array-budget", str(max(64, n // 9 * 8 // 7))]
    if case.get("is_dna"):
        args.append("-d")
    if case.get("allow_ambiguity"):
        args.append("-a")
    if case.get("ignore_softmask"):
        args.append("-i")
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == golden
    assert not (tmp_path / (name + ".partial")).exists()


def test_out_of_core_create_equals_the_whole_array_create(tmp_path):
    """6 Mb of DNA with soft-masked stretches, N runs and repeats over window ends, in windows of 1.1 Mb: budgets that make 1, 3, 8
    and 23 shards write the file the whole-array path writes -- plain, --allow-ambiguity and -m 25; a seed mask keeps the
    whole-array path whatever the budget says"""
    rng = np.random.default_rng(77)
    t = repeat_text(6_000_000, 31, 4000, 60)
    body = t[:-1].copy()
    for _ in range(40):
        p = int(rng.integers(0, body.size - 3000)); body[p : p + int(rng.integers(5, 3000))] = ord("N")
    low = body.copy()
    for _ in range(60):
        p = int(rng.integers(0, low.size - 5000)); q = p + int(rng.integers(50, 5000)); low[p:q] |= 0x20
    fa = tmp_path / "r.fa"
    fa.write_bytes(b">a\n" + low[: low.size // 2].tobytes() + b"\n>b\n" + low[low.size // 2 :].tobytes() + b"\n")
    n = low.size + 2
    for extra in (["-d"], ["-d", "-a"], ["-d", "-m", "25"], ["-d", "-i"], ["-d", "-s", "1101"]):
        ref = tmp_path / "ref.sufr"
        r = subprocess.run([str(sufr_amd.CLI_PATH), "create", str(fa), "-o", str(ref), "--window", "1100000", "--margin", "5000"] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        want = ref.read_bytes()
        for shards, devs in ((1, "0"), (3, "0"), (8, "0,0,0"), (23, "0,0")):
            out = tmp_path / "o.sufr"
            r = subprocess.run([str(sufr_amd.CLI_PATH), "--devices", devs, "create", str(fa), "-o", str(out), "--window", "1100000", "--margin", "5000",
                                "--array-budget", str(2 * n * 4 // shards + 8)] + extra, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            assert out.read_bytes() == want, (extra, shards, devs)


def test_out_of_core_create_through_the_c_abi(tmp_path):
    """sufr_hip_set_array_budget + sufr_hip_create_file on one context: the same file; budget 0 restores the whole-array path"""
    import ctypes as C
    from sufr_amd import _lib, cli
    t = repeat_text(300_000, 5, 700, 30)
    fa = tmp_path / "r.fa"
    fa.write_bytes(b">r\n" + t[:-1].tobytes() + b"\n")
    files = []
    ctx = _lib.Context(0)
    ctx.set_window(50_000, 2_000)
    for budget in (0, 300_000, 0):
        ctx.set_array_budget(budget)
        out = tmp_path / f"o{len(files)}.sufr"
        a = cli.create_args(str(fa), str(out), is_dna=True)
        path = C.create_string_buffer(4096)
        st = _lib.Stats()
        ctx.check(_lib.lib().sufr_hip_create_file(ctx.handle, C.byref(a), path, len(path), C.byref(st)))
        files.append(out.read_bytes())
    ctx.close()
    assert files[0] == files[1] == files[2]


def test_skewed_shard_grows_its_window_rank_arrays():
    """A shard's window ranks get a shard's share of memory (n / K x 1.5 + 2^20 entries) and grow, contents kept, when a window needs
    more.  5.2 M 'A' followed by random DNA, five shards in windows of 1.3 M: every suffix of the run starts with the same 8 bytes,
    so ONE shard holds 5.2 M suffixes against room for 2.8 M -- its arrays grow twice while windows are being filtered.  The shards
    concatenate to the one-window build of the same text."""
    rng = np.random.default_rng(5)
    n = 6_000_000
    t = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
    t[100_000:5_300_000] = ord("A")
    t[-1] = ord("$")
    x = torch.from_numpy(t).cuda()
    db = sufr_amd.DeviceBuilder(0)
    one_sa, one_lcp = db.sort(x, is_dna=True)
    one_sa = one_sa.cpu().numpy().view(np.uint32).astype(np.uint64); one_lcp = one_lcp.cpu().numpy().view(np.uint32).astype(np.uint64)
    db.close()
    sa, lcp, sizes, _ = sharded_windowed(x, n, 5, 1_300_000, 4096, 8, is_dna=True)
    assert max(sizes) > 5_000_000 and sum(sizes) == one_sa.size
    assert np.array_equal(sa, one_sa) and np.array_equal(lcp, one_lcp)


# ---- the one-rank-per-GPU file writer (sufr_hip_shard_build / sufr_write_frame / sufr_hip_shard_write) on texts in windows ----

def _rank_style_create(fasta, out, num_shards, window, margin, **flags):
    """what N ranks do, one after the other in this process: every 'rank' builds its shard on its own context (which keeps it),
    the {first, last, count} triples are exchanged, rank 0 writes the frame, every rank streams its slice"""
    import ctypes as C
    from sufr_amd import _lib, cli, shards
    L = _lib.lib()
    sd = _lib.SequenceData()
    err = C.create_string_buffer(512)
    assert L.sufr_read_sequence_file(str(fasta).encode(), ord("%"), C.byref(sd), err, len(err)) == 0, err.value
    args = cli.create_args(str(fasta), str(out), **flags)
    ctxs, infos = [], []
    try:
        for r in range(num_shards):
            ctx = _lib.Context(0)
            ctx.set_window(window, margin)
            info = _lib.ShardInfo(); st = _lib.Stats()
            ctx.check(L.sufr_hip_shard_build(ctx.handle, C.byref(sd), C.byref(args), r, num_shards, C.byref(info), C.byref(st)))
            ctxs.append(ctx); infos.append(info)
        bounds = [(int(i.first_suffix), int(i.last_suffix), int(i.num_suffixes)) for i in infos]
        total = sum(b[2] for b in bounds)
        assert L.sufr_write_frame(str(out).encode(), C.byref(sd), C.byref(args), total, err, len(err)) == 0, err.value
        for r in reversed(range(num_shards)):                # (any order: every slice has its own place)
            offset, tot, has_prev, prev_last = shards.write_plan(bounds, r)
            assert tot == total
            ctxs[r].check(L.sufr_hip_shard_write(ctxs[r].handle, C.byref(sd), C.byref(args), str(out).encode(), bounds[r][2], total, offset,
                                                 int(has_prev), prev_last, int(r == 0)))
        # a write consumes the shard: a second one has nothing to write
        rc = L.sufr_hip_shard_write(ctxs[0].handle, C.byref(sd), C.byref(args), str(out).encode(), bounds[0][2], total, 0, 0, 0, 0)
        assert rc == -1 and b"no windowed shard" in L.sufr_hip_last_error(ctxs[0].handle)    # SUFR_HIP_E_INVALID
        return [b[2] for b in bounds]
    finally:
        for c in ctxs:
            c.close()
        L.sufr_sequence_data_free(C.byref(sd))


@pytest.mark.parametrize("name", ["long_dna_sequence.sufr", "long_dna_sequence_allow_ambiguity.sufr", "uniprot.sufr", "3.sufr"])
def test_rank_style_create_in_windows_writes_the_golden_file(tmp_path, name):
    """rounds 2-4 (and round 5 until its last day) refused sufr_hip_shard_build on texts that take windows: the ranks of a
    torch.distributed job could not write such an index.  Three ranks, windows of a fortieth of the file: the golden file."""
    case = GOLDEN_CASES[name]
    golden = (GOLDEN / "expected" / name).read_bytes()
    out = tmp_path / name
    flags = {k: True for k in ("is_dna", "allow_ambiguity", "ignore_softmask") if case.get(k)}
    sizes = _rank_style_create(GOLDEN / "inputs" / case["fa"], out, 3, max(16, len(golden) // 40), 48, **flags)
    assert len(sizes) == 3
    assert out.read_bytes() == golden


def test_rank_style_create_in_windows_equals_the_one_context_create(tmp_path):
    """a 60 kb text with repeats across the window ends, five ranks, plain and -m 40: the file `sufr create --window` writes"""
    t = repeat_text(60_000, 9, 900, 40)
    fa = tmp_path / "r.fa"
    fa.write_bytes(b">r\n" + t[:-1].tobytes() + b"\n")
    for extra, flags in (([], {}), (["-m", "40"], {"max_query_len": 40})):
        ref = tmp_path / "ref.sufr"
        r = subprocess.run([str(sufr_amd.CLI_PATH), "create", "-d", str(fa), "-o", str(ref), "--window", "7000", "--margin", "300"] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        out = tmp_path / "o.sufr"
        _rank_style_create(fa, out, 5, 7000, 300, is_dna=True, **flags)
        assert out.read_bytes() == ref.read_bytes()
