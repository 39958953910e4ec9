"""CPU checks of the run-key format shared with the device code (sufr_amd/csrc/sufr_runkey.h):
for suffixes that agree on their first d characters, the integer order of make_run_key(idx + d) must equal
the suffix order whenever the keys differ, run_key_common must give the exact number of further common
characters, and equal keys must mean that run_key_advance(.., 64) further characters agree."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
PAD = 4096 + 256


@pytest.fixture(scope="module")
def shim():
    out = ROOT / "tests" / "_build" / "librunkey_shim.so"
    out.parent.mkdir(exist_ok=True)
    src = ROOT / "tests" / "runkey_shim.cpp"
    hdr = ROOT / "sufr_amd" / "csrc" / "sufr_runkey.h"
    if not out.exists() or out.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(out), str(src)], check=True)
    L = C.CDLL(str(out))
    L.shim_run_lengths.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.shim_run_table_packed.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.shim_run_len_at.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]; L.shim_run_len_at.restype = C.c_uint32
    L.shim_make_run_key.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p]
    L.shim_pack_codes.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
    L.shim_make_run_key.restype = C.c_uint64
    L.shim_run_key_common.argtypes = [C.c_uint64, C.c_uint64, C.c_int]; L.shim_run_key_common.restype = C.c_uint32
    L.shim_run_key_advance.argtypes = [C.c_uint64, C.c_int, C.c_int]; L.shim_run_key_advance.restype = C.c_uint32
    return L


def make_lut(text):
    present = sorted(set(text.tolist()))
    lut = np.zeros(256, dtype=np.uint16)
    for i, c in enumerate(present):
        lut[c] = i + 1
    bits = 1
    while (1 << bits) <= len(present):
        bits += 1
    return lut, bits


def true_lcp(t, n, a, b):
    k = 0
    while a + k < n and b + k < n and t[a + k] == t[b + k]:
        k += 1
    return k


def texts():
    rng = np.random.default_rng(0)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    t = acgt[rng.integers(0, 4, size=3000)].copy()
    for s in range(0, 3000, 300):                      # N runs and homopolymers of many lengths
        ln = int(rng.integers(1, 120)); t[s:s + ln] = ord("N")
        ln2 = int(rng.integers(1, 60)); t[s + 150:s + 150 + ln2] = acgt[int(rng.integers(0, 4))]
    out.append(np.concatenate([t, np.frombuffer(b"$", dtype=np.uint8)]))
    t = acgt[rng.integers(0, 4, size=4000)].copy()     # planted near-identical copies (1 % divergence)
    fam = acgt[rng.integers(0, 4, size=400)]
    for k in range(8):
        c = fam.copy(); m = rng.integers(0, 400, size=4); c[m] = acgt[rng.integers(0, 4, size=4)]
        t[k * 450:k * 450 + 400] = c
    out.append(np.concatenate([t, np.frombuffer(b"$", dtype=np.uint8)]))
    t = acgt[rng.integers(0, 4, size=2500)].copy()     # tandem arrays, unit 1..6
    for u in range(1, 7):
        t[u * 350:u * 350 + 200] = np.resize(acgt[rng.integers(0, 4, size=u)], 200)
    out.append(np.concatenate([t, np.frombuffer(b"$", dtype=np.uint8)]))
    out.append(np.frombuffer(b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA$", dtype=np.uint8).copy())
    out.append(np.frombuffer(b"ABABABABABABABABABABABABABABABABABABABABABAB", dtype=np.uint8).copy())   # no sentinel
    return out


@pytest.mark.parametrize("ti", range(5))
@pytest.mark.parametrize("pi", [1, 2, 3, 5])
def test_run_keys_are_order_preserving_and_decode_exact_lcp(shim, ti, pi):
    t = texts()[ti]
    n = t.size
    buf = np.zeros(n + PAD, dtype=np.uint8); buf[:n] = t
    R = np.zeros((n // 4096 + 2) * 66, dtype=np.uint64)     # run-end tables (RunTable of sufr_runkey.h)
    shim.shim_run_table_packed(buf.ctypes.data, n, R.ctypes.data)
    lut, bits = make_lut(t)
    packed = np.zeros(n * bits // 8 + 128, dtype=np.uint8)
    shim.shim_pack_codes(buf.ctypes.data, n, lut.ctypes.data, bits, packed.ctypes.data, packed.size)
    rng = np.random.default_rng(ti * 10 + pi)
    tb = t.tobytes()
    checked = 0
    # pairs with a common prefix: sort suffixes, take neighbours at small distance in the suffix array
    order = sorted(range(n), key=lambda i: tb[i:])
    for _ in range(4000):
        r = int(rng.integers(0, n - 1)); s = min(n - 1, r + int(rng.integers(1, 6)))
        a, b = order[r], order[s]                 # suffix a < suffix b
        l = true_lcp(tb, n, a, b)
        if l < pi:
            continue
        d = int(rng.integers(pi, l + 1))          # any depth inside the common prefix (>= pi)
        ka = shim.shim_make_run_key(buf.ctypes.data, n, R.ctypes.data, lut.ctypes.data, bits, a + d, pi, None)
        kb = shim.shim_make_run_key(buf.ctypes.data, n, R.ctypes.data, lut.ctypes.data, bits, b + d, pi, None)
        # the bit-packed fast path must produce the very same keys
        assert ka == shim.shim_make_run_key(buf.ctypes.data, n, R.ctypes.data, lut.ctypes.data, bits, a + d, pi,
                                            packed.ctypes.data)
        assert kb == shim.shim_make_run_key(buf.ctypes.data, n, R.ctypes.data, lut.ctypes.data, bits, b + d, pi,
                                            packed.ctypes.data)
        rest = l - d
        if ka == kb:
            adv = shim.shim_run_key_advance(ka, 64, bits)
            assert adv <= rest, (a, b, d, adv, rest)
        else:
            assert ka < kb, f"order: a={a} b={b} d={d} l={l} ka={ka:016x} kb={kb:016x}"
            com = shim.shim_run_key_common(ka, kb, bits)
            assert com == rest, f"common: a={a} b={b} d={d} l={l} got {com} want {rest} ka={ka:016x} kb={kb:016x}"
        checked += 1
    assert checked > 200


def test_run_len_at_matches_bytewise_runs(shim):
    """run_len_at (run-end bitmap, per-tile summary word, per-tile first end) == the byte-at-a-time definition,
    on runs that end inside a tile, at a tile edge, many tiles later, beyond the saturation and at the end of
    the text (zero bytes included: the padding after the text is zero too)."""
    rng = np.random.default_rng(3)
    parts = []
    for ln in [1, 2, 7, 8, 9, 63, 4095, 4096, 4097, 3, 12_000, 5, 70_000, 1, 140_000, 2, 66_000, 4, 3_000_000]:
        parts.append(np.full(ln, rng.integers(0, 4), dtype=np.uint8))
        parts.append(rng.integers(4, 9, size=int(rng.integers(1, 40)), dtype=np.uint8))
    for tail in [np.empty(0, np.uint8), np.zeros(9000, np.uint8), np.full(5000, 65, np.uint8)]:
        t = np.concatenate(parts + [tail])
        # adjacent equal values across part boundaries merge into longer runs: that is fine
        n = t.size
        buf = np.zeros(n + PAD, dtype=np.uint8); buf[:n] = t
        want = np.zeros(n + PAD, dtype=np.uint32)
        shim.shim_run_lengths(buf.ctypes.data, n, want.ctypes.data)
        tab = np.zeros((n // 4096 + 2) * 66, dtype=np.uint64)
        shim.shim_run_table_packed(buf.ctypes.data, n, tab.ctypes.data)
        qs = np.unique(np.concatenate([rng.integers(0, n, 30_000), np.arange(max(0, n - 70_000), n),
                                       np.arange(0, min(n, 9000))]))
        for q in qs.tolist():
            got = shim.shim_run_len_at(n, q, tab.ctypes.data)
            assert got == int(want[q]), (q, got, int(want[q]))


def test_dna3_digest_is_monotone_and_spreads_acgt(shim):
    """The counting digit of the leaf sort for 3-bit DNA keys (k_leaf_sort<.., DNA3>): monotone over all 2^18
    six-character strings (records are bucketed by it and only ranked by whole keys INSIDE a bucket), 2 bits per
    A / C / G / T, and nothing after a pad / '$' / '%' / N counts."""
    shim.shim_dna3_digest_all.argtypes = [C.c_void_p]
    d = np.zeros(1 << 18, dtype=np.uint32)
    shim.shim_dna3_digest_all(d.ctypes.data)
    assert (np.diff(d.astype(np.int64)) >= 0).all()
    t = np.arange(1 << 18, dtype=np.uint32)
    ref = np.zeros_like(t)
    cut = np.zeros(t.shape, dtype=bool)
    for i in range(5, -1, -1):
        c = (t >> (3 * i)) & 7
        m = np.select([c < 4, c == 4, c == 5], [0, 1, 2], 3).astype(np.uint32)
        ref = (ref << 2) | np.where(cut, 0, m).astype(np.uint32)
        cut |= (c < 3) | (c == 6)
    assert np.array_equal(d, ref)
    acgt = np.array([3, 4, 5, 7])
    idx = np.zeros(4 ** 6, dtype=np.uint32)
    for i in range(6):
        idx = (idx << 3) | acgt[(np.arange(4 ** 6) >> (2 * (5 - i))) & 3].astype(np.uint32)
    assert np.array_equal(d[idx], np.arange(4 ** 6, dtype=np.uint32))       # a bijection on ACGT^6
