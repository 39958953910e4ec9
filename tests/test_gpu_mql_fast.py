"""--max-query-len builds whose cap fits the 64-bit key (DNA: 11 <= L <= 21; sufr_launch.inc "mql_fast", round 6): the capped
build is built directly -- keys of L characters + position bits, ties by descending position -- instead of the exact build
followed by apply_max_query_len.  Expected arrays: the canonical member of the reference's family,
computed here from the ORACLE's exact arrays: LCP = min(LCP, L), every run of LCP >= L in descending position
(sufr_builder.rs:310-314, 668-712).  Whole-array equality."""
import os

import numpy as np
import pytest
import torch

import sufr_amd
from sufr_amd import synth

pytestmark = pytest.mark.gpu


def canonical(osa: np.ndarray, olcp: np.ndarray, L: int):
    """the exact arrays -> the capped build's: runs of ranks with LCP >= L re-ordered by descending position"""
    lcp = np.minimum(olcp, L).astype(np.uint32)
    head = np.ones(osa.size, dtype=bool)
    head[1:] = olcp[1:] < L
    run = np.cumsum(head) - 1
    order = np.lexsort((-osa.astype(np.int64), run))
    return osa[order], lcp


def _acgt(rng, n, p=None):
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, n, p=p)].copy()


def _check(oracle, raw, L, **kw):
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    osa, olcp, _ = oracle.build(oracle.normalize(raw, kw.get("ignore_softmask", False)), is_dna=True,
                                allow_ambiguity=kw.get("allow_ambiguity", False), threads=min(16, os.cpu_count() or 1))
    wsa, wlcp = canonical(osa, olcp, L)
    db = sufr_amd.DeviceBuilder(0)
    sa, lcp = db.sort(torch.from_numpy(raw).cuda(), raw_text=True, is_dna=True, max_query_len=L, **kw)
    gsa = sa.cpu().numpy().view(np.uint32); glcp = lcp.cpu().numpy().view(np.uint32)
    st = db.stats
    db.close()
    bad = np.nonzero(glcp != wlcp)[0]
    assert bad.size == 0, f"L={L}: LCP differs at rank {bad[0]} of {wsa.size}: got {glcp[bad[0]]} want {wlcp[bad[0]]} ({bad.size} ranks)"
    bad = np.nonzero(gsa != wsa)[0]
    assert bad.size == 0, f"L={L}: SA differs at rank {bad[0]} of {wsa.size}: got {gsa[bad[0]]} want {wsa[bad[0]]} ({bad.size} ranks)"
    return st


@pytest.mark.parametrize("L", [11, 12, 16, 20, 21])
@pytest.mark.parametrize("n", [300, 50_000, 2_000_000])
def test_random_and_low_entropy_dna(oracle, n, L):
    rng = np.random.default_rng(n + L)
    raw = _acgt(rng, n, p=[0.55, 0.15, 0.15, 0.15])
    raw[-1] = ord("$")
    st = _check(oracle, raw, L)
    assert st.chars_per_key == L                      # the capped build proper, not exact build + re-ordering


@pytest.mark.parametrize("L", [11, 12, 16, 21])
def test_large_groups_of_equal_prefixes(oracle, L):
    """what the position bits below the characters cannot tell apart: a homopolymer run of 300 000 (under -m 16 its suffixes
    tie in 64 K blocks of positions: tie runs far above a finisher window), 20 000 copies of a 40-mer spread over the text
    (an L-group of 20 000: MSD levels on the position bits, left-over groups under -m 21), tandem arrays, and N runs with
    --allow-ambiguity"""
    rng = np.random.default_rng(L)
    raw = _acgt(rng, 3_000_000)
    raw[100_000:400_000] = ord("A")
    unit = _acgt(rng, 40)
    at = rng.choice(np.arange(500_000, 2_900_000, 64), 20_000, replace=False)
    raw[at[:, None] + np.arange(40)[None, :]] = unit[None, :]
    raw[450_000:470_000] = np.resize(np.frombuffer(b"ACG", dtype=np.uint8), 20_000)
    raw[2_950_000:2_990_000] = ord("N")
    raw[-1] = ord("$")
    _check(oracle, raw, L)
    _check(oracle, raw, L, allow_ambiguity=True)


@pytest.mark.parametrize("L", [12, 16])
def test_genome_shaped_text_softmasked_repeats(oracle, L):
    x, _ = synth.syn_human(4_000_000, seed=23)
    raw = x.numpy()
    _check(oracle, raw, L)
    _check(oracle, raw, L, ignore_softmask=True)


@pytest.mark.parametrize("L,shards", [(12, 3), (16, 5), (21, 2)])
def test_shards_and_windows_of_a_capped_build_concatenate(oracle, L, shards):
    rng = np.random.default_rng(L * shards)
    raw = _acgt(rng, 600_000, p=[0.4, 0.2, 0.2, 0.2])
    raw[200_000:260_000] = ord("T")
    raw[300_000:303_000] = raw[100_000:103_000]
    raw[-1] = ord("$")
    osa, olcp, _ = oracle.build(raw, is_dna=True, threads=8)
    wsa, wlcp = canonical(osa, olcp, L)
    x = torch.from_numpy(raw).cuda()
    db = sufr_amd.DeviceBuilder(0)
    parts_sa, parts_lcp = [], []
    for k in range(shards):
        sa, lcp = db.sort(x, raw_text=True, is_dna=True, max_query_len=L, shard_index=k, num_shards=shards)
        parts_sa.append(sa.cpu().numpy().view(np.uint32).copy()); parts_lcp.append(lcp.cpu().numpy().view(np.uint32).copy())
    gsa = np.concatenate(parts_sa); glcp = np.concatenate(parts_lcp)
    assert np.array_equal(gsa, wsa)
    starts = np.cumsum([0] + [p.size for p in parts_sa[:-1]])
    keep = np.ones(wsa.size, dtype=bool); keep[starts[1:]] = False      # (a shard's first LCP is the stitch's)
    assert np.array_equal(glcp[keep], wlcp[keep])
    # the same text in four 32-bit windows (the path of texts beyond 2^32 bytes)
    db.ctx.set_window(160_000, 70_000)
    sa8, lcp8 = db.sort(x, raw_text=True, is_dna=True, max_query_len=L, index_width=8)
    assert np.array_equal(sa8.cpu().numpy().astype(np.uint32), wsa) and np.array_equal(lcp8.cpu().numpy().astype(np.uint32), wlcp)
    db.ctx.set_window(0, 0)
    db.close()


def test_caps_outside_the_key_keep_the_two_step_form(oracle):
    """L < 11 (the characters do not fill the key's high word) and L > 21: exact build + re-ordering, same canonical arrays"""
    rng = np.random.default_rng(77)
    raw = _acgt(rng, 200_000, p=[0.5, 0.2, 0.2, 0.1])
    raw[-1] = ord("$")
    for L in (3, 10, 22, 60):
        st = _check(oracle, raw, L)
        assert st.chars_per_key == 21


@pytest.mark.parametrize("L,shards", [(11, 1), (12, 1), (16, 3), (21, 2)])
@pytest.mark.parametrize("amb", [False, True])
def test_capped_builds_with_bytes_outside_the_dna_table(oracle, L, shards, amb):
    """a real assembly's IUPAC letters must not take a `-m` build off the direct path either: the build runs on 'N' for the listed
    bytes, and the suffixes whose first L symbols hold one are re-placed under the CAPPED order (first L true bytes, ties in
    descending position).  Letters inside copies of a repeat (equal L-prefixes with and without the letter), next to each other,
    in a homopolymer run; one shard and several."""
    rng = np.random.default_rng(100 * L + shards + amb)
    raw = _acgt(rng, 700_000, p=[0.4, 0.2, 0.2, 0.2])
    seg = _acgt(rng, 300)
    for i, c in enumerate(b"RNTAYRK-"):
        s_ = seg.copy(); s_[150] = c
        raw[50_000 + i * 70_000:50_000 + i * 70_000 + 300] = s_
    raw[600_000:640_000] = ord("A"); raw[620_000] = ord("W"); raw[620_001] = ord("S")
    raw[rng.integers(0, raw.size - 1, 40)] = np.frombuffer(b"RYKMSWBDHV#", dtype=np.uint8)[rng.integers(0, 11, 40)]
    raw[-1] = ord("$")
    osa, olcp, _ = oracle.build(raw, is_dna=True, allow_ambiguity=amb, threads=8)
    wsa, wlcp = canonical(osa, olcp, L)
    x = torch.from_numpy(raw).cuda()
    db = sufr_amd.DeviceBuilder(0)
    sas, lcps = [], []
    for k in range(shards):
        sa, lcp = db.sort(x, raw_text=True, is_dna=True, allow_ambiguity=amb, max_query_len=L, shard_index=k, num_shards=shards)
        assert db.stats.num_exceptions > 0 and db.stats.chars_per_key == L and db.stats.bits_per_char == 3
        sas.append(sa.cpu().numpy().view(np.uint32).copy()); lcps.append(lcp.cpu().numpy().view(np.uint32).copy())
    db.close()
    gsa = np.concatenate(sas); glcp = np.concatenate(lcps)
    assert gsa.size == wsa.size
    bad = np.nonzero(gsa != wsa)[0]
    assert bad.size == 0, f"SA differs at rank {bad[0]}: got {gsa[bad[0]]} want {wsa[bad[0]]} ({bad.size} ranks)"
    starts = np.cumsum([0] + [p.size for p in sas[:-1]])
    keep = np.ones(wsa.size, dtype=bool); keep[starts[1:]] = False
    bad = np.nonzero((glcp != wlcp) & keep)[0]
    assert bad.size == 0, f"LCP differs at rank {bad[0]}: got {glcp[bad[0]]} want {wlcp[bad[0]]} ({bad.size} ranks)"

