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
from sufr_amd import synth, verify
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


def test_masked_and_truncated_builds_are_refused_in_windows():
    t = torch.from_numpy(repeat_text(20_000, 4, 100, 5)).cuda()
    db = sufr_amd.DeviceBuilder(0)
    db.ctx.set_window(4096, 64)
    with pytest.raises(sufr_amd.SufrHipError) as e:
        db.sort(t, is_dna=True, max_query_len=8)
    assert e.value.code == -6 and "one 32-bit window" in e.value.message
    with pytest.raises(sufr_amd.SufrHipError):
        db.sort(t, is_dna=True, seed_mask="1101")
    db.ctx.set_window(0, 0)                                   # back to one window: the same context builds them
    sa, _ = db.sort(t, is_dna=True, max_query_len=8)
    assert sa.numel() > 0
    db.close()


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
    db.close()
    del x, sa, lcp
    torch.cuda.empty_cache()


def test_cli_refuses_masked_builds_in_windows_before_building(tmp_path):
    fa = GOLDEN / "inputs" / "long_dna_sequence.fa"
    for extra in (["-m", "8"], ["-s", "1101"]):
        r = subprocess.run([str(sufr_amd.CLI_PATH), "create", str(fa), "-d", "-o", str(tmp_path / "x.sufr"), "--window", "4096", *extra],
                           capture_output=True, text=True)
        assert r.returncode == 1 and "one 32-bit window" in r.stderr
        assert not (tmp_path / "x.sufr").exists()
    r = subprocess.run([str(sufr_amd.CLI_PATH), "create", str(fa), "-d", "-o", str(tmp_path / "y.sufr"), "--window", "4096", "-m", "0"],
                       capture_output=True, text=True)                                   # Some(0): a plain build
    assert r.returncode == 0, r.stderr
