"""Batched search on the GPU (include/sufr_query.h, device section) against the host search of the same library and
against properties of the answer that do not depend on either: every rank inside the range matches the query, the ranks
on both sides do not."""
import ctypes as C
import zlib

import numpy as np
import pytest
import torch

import sufr_amd
from sufr_amd import DeviceIndex, SufrFile, pack_queries
from oracle_helper import GOLDEN
from test_query import random_queries, witness_ranks

pytestmark = pytest.mark.gpu
EXP = GOLDEN / "expected"


@pytest.fixture(scope="module")
def ctx():
    c = sufr_amd.Context(0)
    yield c
    c.close()


def host_ranges(f, queries, mql):
    lo = np.zeros(len(queries), dtype=np.uint64)
    hi = np.zeros(len(queries), dtype=np.uint64)
    for i, q in enumerate(queries):
        r = f.search(q, mql)
        if r:
            lo[i], hi[i] = r
    return lo, hi


@pytest.mark.parametrize("name", sorted(p.name for p in EXP.glob("*.sufr")))
@pytest.mark.parametrize("mql", [None, 1, 3, 6])
def test_device_search_equals_host_search_on_golden_files(ctx, name, mql):
    f = SufrFile(EXP / name)
    ix = DeviceIndex.load(ctx, f)
    rng = np.random.default_rng(zlib.crc32(f"{name}{mql}".encode()))
    max_len = len(f.seed_mask) if f.seed_mask else 16
    queries = random_queries(rng, f, 3000, max_len)
    lo, hi = ix.search(queries, mql)
    wlo, whi = host_ranges(f, queries, mql)
    assert np.array_equal(lo, wlo) and np.array_equal(hi, whi)
    assert (hi > lo).sum() > 0
    if f.text_len < 3000:                                   # and the scan of every rank, where that is cheap
        for i in range(0, len(queries), 37):
            want = witness_ranks(f, queries[i], mql)
            assert (int(lo[i]), int(hi[i])) == ((want[0], want[-1] + 1) if want else (0, 0))
    ix.close()


def test_reference_cli_vectors_on_the_device(ctx):          # sufr/tests/cli.rs:339-360, 949-1060
    f = SufrFile(EXP / "1.sufr")
    assert [c.count for c in DeviceIndex.load(ctx, f).count(["AC", "X", "GT"])] == [2, 0, 2]
    f = SufrFile(EXP / "uniprot-masked.sufr")
    lo, hi = DeviceIndex.load(ctx, f).search(["RNEL"])
    assert f.suffix_array[int(lo[0]):int(hi[0])].tolist() == [54791, 46515, 37970, 62005, 52278, 6386, 4124]
    f = SufrFile(EXP / "long_dna_sequence.sufr")
    lo, hi = DeviceIndex.load(ctx, f).search(["CATGTTGTCACG", "CCATGGGAC", "GGATGAAGAAAAGCA"], 6)
    assert [sorted(f.suffix_array[int(a):int(b)].tolist()) for a, b in zip(lo, hi)] == \
        [[2566, 13056, 20444], [3014], [1026, 2905, 13253, 13508, 14250, 14624, 20465]]


def test_empty_batch_and_empty_query(ctx):
    f = SufrFile(EXP / "1.sufr")
    ix = DeviceIndex.load(ctx, f)
    lo, hi = ix.search([])
    assert lo.size == 0 and hi.size == 0
    lo, hi = ix.search([b"", b"AC"])
    assert (lo.tolist(), hi.tolist()) == ([0, 1], [9, 3])


def check_range_properties(text, sa, qbytes, off, lo, hi, sample):
    """text, sa: numpy; for the sampled queries: SA[lo], SA[hi-1] start with the query, SA[lo-1] and SA[hi] do not,
    and when the query does not occur the text does not contain it (checked by bytes.find on a window is too slow at this
    size: the miss side is covered by comparing with the host search)."""
    n, s = text.size, sa.size
    tb = text.tobytes()

    def starts_with(pos, q):
        return tb[pos:pos + len(q)] == q

    for i in sample:
        q = qbytes[int(off[i]):int(off[i + 1])].tobytes()
        a, b = int(lo[i]), int(hi[i])
        if b > a:
            assert starts_with(int(sa[a]), q) and starts_with(int(sa[b - 1]), q)
            assert a == 0 or not starts_with(int(sa[a - 1]), q)
            assert b == s or not starts_with(int(sa[b]), q)
            mid = (a + b) // 2
            assert starts_with(int(sa[mid]), q)


def test_build_then_query_without_leaving_the_device(ctx, tmp_path):
    """20 Mb of DNA with planted repeats: built by the device builder, wrapped in place, 200 000 queries; the same file
    written to disk and searched by the host code gives the same ranges."""
    dev = "cuda"
    g = torch.Generator(device=dev); g.manual_seed(3)
    n = 20_000_001
    x = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n,), generator=g, device=dev)]
    x[5_000_000:5_200_000] = x[1_000_000:1_200_000]
    x[9_000_000:9_000_500] = ord("A")
    x[-1] = ord("$")
    db = sufr_amd.DeviceBuilder(0)
    sa, lcp = db.sort(x, is_dna=True)
    ix = DeviceIndex.wrap(db.ctx, x, sa)
    text = x.cpu().numpy()
    sa_h = sa.cpu().numpy().view(np.uint32)
    rng = np.random.default_rng(9)
    nq = 200_000
    lens = rng.integers(4, 41, nq)
    at = rng.integers(0, n - 64, nq)
    off = np.zeros(nq + 1, dtype=np.uint64); off[1:] = np.cumsum(lens)
    qb = np.empty(int(off[-1]), dtype=np.uint8)
    for i in range(nq):
        qb[int(off[i]):int(off[i + 1])] = text[at[i]:at[i] + lens[i]]
    flip = rng.integers(0, qb.size, nq // 3)                  # a third of the queries get one changed symbol
    qb[flip] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, flip.size)]
    lo, hi = ix.search_packed(qb, off)
    found = hi > lo
    assert 0.6 < found.mean() < 1.0
    assert int((hi - lo).max()) > 400                           # the poly-A run
    check_range_properties(text, sa_h, qb, off, lo, hi, rng.integers(0, nq, 20_000))
    # the prefix table (k = 13 here) changes no answer: the same batch without it, with -m below / at / above k, and
    # queries that start with symbols outside the table alphabet or are shorter than k
    plain = DeviceIndex.wrap(db.ctx, x, sa, prefix_table=False)
    plo, phi = plain.search_packed(qb, off)
    assert np.array_equal(plo, lo) and np.array_equal(phi, hi)
    odd = qb.copy()
    odd[rng.integers(0, odd.size, nq // 2)] = ord("N")
    short_off = np.zeros(nq + 1, dtype=np.uint64); short_off[1:] = np.cumsum(np.minimum(lens, rng.integers(1, 16, nq)))
    for qbytes, offsets, mql in ((odd, off, None), (qb, short_off, None), (qb, off, 5), (qb, off, 13), (qb, off, 20)):
        a = ix.search_packed(qbytes[:int(offsets[-1])], offsets, mql)
        b = plain.search_packed(qbytes[:int(offsets[-1])], offsets, mql)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), mql
    plain.close()
    # device buffers in, device buffers out
    dlo, dhi = ix.search_device(torch.from_numpy(qb).to(dev), torch.from_numpy(off.astype(np.int64)).to(dev))
    assert np.array_equal(dlo.cpu().numpy().astype(np.uint64), lo) and np.array_equal(dhi.cpu().numpy().astype(np.uint64), hi)
    # the host search of the written file
    path = tmp_path / "x.sufr"
    lcp_h = lcp.cpu().numpy().view(np.uint32)
    err = C.create_string_buffer(256)
    starts = np.zeros(1, dtype=np.uint64)
    names = (C.c_char_p * 1)(b"1")
    rc = sufr_amd.lib().sufr_write_file(str(path).encode(), 1, 0, 0, text.ctypes.data, n, 4, sa_h.ctypes.data, lcp_h.ctypes.data,
                                        sa_h.size, 0, 0, None, starts.ctypes.data, 1, names, err, len(err))
    assert rc == 0, err.value
    f = SufrFile(path)
    sub = rng.integers(0, nq, 20_000)
    for i in sub:
        r = f.search(qb[int(off[i]):int(off[i + 1])].tobytes())
        assert (int(lo[i]), int(hi[i])) == (r if r else (0, 0))
    ix.close()
    db.close()


def test_index_of_another_type_or_device_is_refused(ctx):
    x = torch.zeros(16, dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):                       # (64-bit arrays are taken since round 3; 16-bit ones are not)
        DeviceIndex.wrap(ctx, x, torch.zeros(4, dtype=torch.int16, device="cuda"))
    with pytest.raises(ValueError):
        DeviceIndex.wrap(ctx, x.cpu(), torch.zeros(4, dtype=torch.int32, device="cuda"))


# the reference's CLI expectations once more, searched as a batch on the GPU (`--device 0`)
from test_query import EXTRACT_CASES, LOCATE_CASES, run


def test_cli_count_on_the_device():
    assert run("count", "--device", 0, EXP / "1.sufr", "AC", "X", "GT").stdout == "AC 2\nX 0\nGT 2\n"


@pytest.mark.parametrize("case", range(len(LOCATE_CASES)))
def test_cli_locate_on_the_device(case):
    path, queries, opts, expected = LOCATE_CASES[case]
    assert run("locate", path, "--device", 0, *opts, *queries).stdout == (EXP / expected).read_text()


@pytest.mark.parametrize("case", range(len(EXTRACT_CASES)))
def test_cli_extract_on_the_device(case):
    path, queries, opts, lines, err = EXTRACT_CASES[case]
    r = run("extract", path, "--device", 0, *opts, *queries)
    assert r.stdout == "".join(l + "\n" for l in lines) and r.stderr == err


@pytest.mark.parametrize("name", ["abba.sufr", "2.sufr", "long_dna_sequence.sufr", "uniprot.sufr", "uniprot-masked.sufr"])
def test_device_locate_equals_host_locate(ctx, name):
    """positions gathered on the device = SufrFile::locate's suffixes in rank order (sufr_file.rs:1110-1175)"""
    f = SufrFile(EXP / name)
    ix = DeviceIndex.load(ctx, f)
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    queries = random_queries(rng, f, 2000, len(f.seed_mask) if f.seed_mask else 10)
    want = f.locate(queries)
    got = ix.locate(queries)
    assert [g.tolist() for g in got] == [[p.suffix for p in w.positions] for w in want]
    capped = ix.locate(queries, max_hits=3)
    assert [g.tolist() for g in capped] == [[p.suffix for p in w.positions][:3] for w in want]
    assert all(g.size == 0 for g in ix.locate([b"\x01\x02"])) and ix.locate([]) == []
    # room for fewer positions than there are: refused, with the total reported
    qb, off = pack_queries(queries)
    lo, hi = ix.search_device(torch.from_numpy(qb).cuda(), torch.from_numpy(off.astype(np.int64)).cuda())
    total = int((hi - lo).sum())
    with pytest.raises(sufr_amd.SufrHipError) as e:
        ix.locate_device(lo, hi, capacity=total - 1)
    assert e.value.code == -5 and str(total) in e.value.message
    ix.close()


def test_device_search_on_64_bit_suffix_arrays(tmp_path):
    """The u64 arm of the file format (suffix_array.rs:460-470) on the device: the same text built with 32- and with
    64-bit indices gives the same rank ranges and the same positions, from a wrapped array and from a loaded file
    (the reference's u64 KAT input, lib.rs:94-140, written with index width 8 by the windowed build)."""
    import torch
    from sufr_amd import synth
    x, _ = synth.syn_human(400_000, seed=31)
    raw = x.numpy()
    db = sufr_amd.DeviceBuilder(0)
    d_text = torch.from_numpy(raw).cuda()
    sa32, _ = db.sort(d_text, is_dna=True, ignore_softmask=True, raw_text=True)
    sa32 = sa32.clone()
    sa64, _ = db.sort(d_text, is_dna=True, ignore_softmask=True, raw_text=True, index_width=8)
    assert sa64.dtype == torch.int64 and torch.equal(sa64, sa32.to(torch.int64) & 0xFFFFFFFF)
    norm = torch.from_numpy(sufr_amd.normalize(raw, True)).cuda()
    text = norm.cpu().numpy().tobytes()
    rng = np.random.default_rng(5)
    queries = [text[i:i + int(L)] for i, L in zip(rng.integers(0, len(text) - 40, 3000), rng.integers(1, 40, 3000))]
    queries += [b"ACGTTTTTTGGGGGGGGGGCA", b"N" * 30, b"$"]
    a = sufr_amd.DeviceIndex.wrap(db.ctx, norm, sa32, is_dna=True)
    b = sufr_amd.DeviceIndex.wrap(db.ctx, norm, sa64, is_dna=True)
    lo_a, hi_a = a.search(queries)
    lo_b, hi_b = b.search(queries)
    assert np.array_equal(lo_a, lo_b) and np.array_equal(hi_a, hi_b) and int((hi_a > lo_a).sum()) > 1000   # (queries that start inside masked stretches match no suffix start)
    dl = torch.from_numpy(lo_a.astype(np.int64)).cuda(); dh = torch.from_numpy(hi_a.astype(np.int64)).cuda()
    off_a, pos_a = a.locate_device(dl, dh, max_hits=7)
    off_b, pos_b = b.locate_device(dl, dh, max_hits=7)
    assert pos_b.dtype == torch.int64 and torch.equal(off_a, off_b)
    assert torch.equal(pos_b[:int(off_b[-1])], pos_a[:int(off_a[-1])].to(torch.int64) & 0xFFFFFFFF)
    a.close(); b.close(); db.close()
