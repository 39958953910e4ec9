"""ctypes front-end to oracle/ (TEST INFRASTRUCTURE ONLY) plus test-side helpers.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package (sufr_amd) never does.

Contents:
  * Oracle           -- the C restatement of the reference algorithm (oracle/sufr_oracle.c)
  * naive_sa_lcp     -- an independent numpy/Python witness (sorted suffixes + exact LCP)
  * parse_sufr       -- a test-side decoder of the .sufr v6 layout
                        (reference reader: libsufr/src/sufr_file.rs:145-275)
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import subprocess
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
ORACLE_DIR = REPO / "oracle"
GOLDEN = REPO / "tests" / "golden"


class OracleStats(C.Structure):
    _fields_ = [
        ("text_len", C.c_uint64), ("num_suffixes", C.c_uint64),
        ("num_pivots", C.c_uint64), ("num_over_partitions", C.c_uint64),
        ("t_pivots", C.c_double), ("t_partition", C.c_double),
        ("t_sort", C.c_double), ("t_total", C.c_double),
    ]


def build_oracle(native: bool = False) -> Path:
    """Compile oracle/ with gcc (seconds).  Building the checker is not using it."""
    name = "libsufr_oracle_native.so" if native else "libsufr_oracle.so"
    out = ORACLE_DIR / "_build" / name
    srcs = [ORACLE_DIR / "sufr_oracle.c", ORACLE_DIR / "sufr_oracle_body.inc", ORACLE_DIR / "sufr_oracle.h"]
    if out.exists() and all(out.stat().st_mtime >= s.stat().st_mtime for s in srcs):
        return out
    cmd = ["make", "-C", str(ORACLE_DIR)] + (["NATIVE=1"] if native else [])
    subprocess.run(cmd, check=True, capture_output=True)
    return out


class Oracle:
    def __init__(self, native: bool = False):
        self.lib = C.CDLL(str(build_oracle(native)))
        L = self.lib
        L.sufr_oracle_normalize.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        L.sufr_oracle_normalize.restype = None
        L.sufr_oracle_build.argtypes = [
            C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_char_p,
            C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
            C.POINTER(OracleStats), C.c_char_p, C.c_size_t]
        L.sufr_oracle_build.restype = C.c_int
        L.sufr_oracle_kat.argtypes = [
            C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_char_p,
            C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
        L.sufr_oracle_kat.restype = C.c_int64
        L.sufr_oracle_write_file.argtypes = [
            C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_int,
            C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_char_p,
            C.c_void_p, C.c_uint64, C.POINTER(C.c_char_p), C.c_char_p, C.c_size_t]
        L.sufr_oracle_write_file.restype = C.c_int
        L.sufr_oracle_read_sequence_file.argtypes = [
            C.c_char_p, C.c_uint8, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
            C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
            C.c_char_p, C.c_size_t]
        L.sufr_oracle_read_sequence_file.restype = C.c_int
        L.sufr_oracle_free.argtypes = [C.c_void_p]
        L.sufr_oracle_free.restype = None

    # -- text ----------------------------------------------------------
    def normalize(self, raw: bytes | np.ndarray, ignore_softmask: bool) -> np.ndarray:
        a = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw
        a = np.ascontiguousarray(a)
        out = np.empty_like(a)
        self.lib.sufr_oracle_normalize(a.ctypes.data, out.ctypes.data, a.size, int(ignore_softmask))
        return out

    def read_sequence_file(self, path: str | os.PathLike, delimiter: bytes = b"%"):
        seq = C.c_void_p(); seq_len = C.c_uint64(); starts = C.c_void_p(); names = C.c_void_p()
        nseq = C.c_uint64(); err = C.create_string_buffer(512)
        rc = self.lib.sufr_oracle_read_sequence_file(
            str(path).encode(), delimiter[0], C.byref(seq), C.byref(seq_len), C.byref(starts),
            C.byref(names), C.byref(nseq), err, len(err))
        if rc != 0:
            raise RuntimeError(err.value.decode())
        text = bytes((C.c_uint8 * seq_len.value).from_address(seq.value))
        st = list((C.c_uint64 * nseq.value).from_address(starts.value))
        nm_ptrs = (C.c_char_p * nseq.value).from_address(names.value)
        nm = [nm_ptrs[i].decode() for i in range(nseq.value)]
        raw_ptrs = (C.c_void_p * nseq.value).from_address(names.value)
        for i in range(nseq.value):
            self.lib.sufr_oracle_free(raw_ptrs[i])
        self.lib.sufr_oracle_free(seq); self.lib.sufr_oracle_free(starts); self.lib.sufr_oracle_free(names)
        return text, st, nm

    # -- build ---------------------------------------------------------
    def build(self, norm_text: np.ndarray, *, is_dna=False, allow_ambiguity=False,
              max_query_len: int | None = None, seed_mask: str | None = None,
              num_partitions=16, random_seed=42, threads=1, width=4):
        t = np.ascontiguousarray(norm_text, dtype=np.uint8)
        dt = np.uint32 if width == 4 else np.uint64
        sa = np.zeros(t.size, dtype=dt); lcp = np.zeros(t.size, dtype=dt)
        st = OracleStats(); err = C.create_string_buffer(512)
        rc = self.lib.sufr_oracle_build(
            t.ctypes.data, t.size, int(is_dna), int(allow_ambiguity),
            int(max_query_len is not None), int(max_query_len or 0),
            seed_mask.encode() if seed_mask is not None else None,
            num_partitions, random_seed, threads, width, sa.ctypes.data, lcp.ctypes.data,
            C.byref(st), err, len(err))
        if rc != 0:
            raise RuntimeError(err.value.decode())
        s = st.num_suffixes
        return sa[:s].copy(), lcp[:s].copy(), st

    def kat(self, what: str, text: bytes, a=0, b=0, length=0, skip=0, pivots=(),
            max_query_len: int | None = None, seed_mask: str | None = None) -> int:
        t = np.frombuffer(text, dtype=np.uint8)
        pv = np.asarray(list(pivots), dtype=np.uint64)
        return self.lib.sufr_oracle_kat(
            what.encode(), t.ctypes.data, t.size, int(max_query_len is not None),
            int(max_query_len or 0), seed_mask.encode() if seed_mask is not None else None,
            a, b, length, skip, pv.ctypes.data if pv.size else None, pv.size)

    def write_file(self, path, *, is_dna, allow_ambiguity, ignore_softmask, norm_text, sa, lcp,
                   width=4, max_query_len=None, seed_mask=None, sequence_starts=(0,),
                   sequence_names=("1",)):
        t = np.ascontiguousarray(norm_text, dtype=np.uint8)
        dt = np.uint32 if width == 4 else np.uint64
        sa = np.ascontiguousarray(sa, dtype=dt); lcp = np.ascontiguousarray(lcp, dtype=dt)
        starts = np.asarray(list(sequence_starts), dtype=np.uint64)
        names = (C.c_char_p * len(sequence_names))(*[n.encode() for n in sequence_names])
        err = C.create_string_buffer(512)
        rc = self.lib.sufr_oracle_write_file(
            str(path).encode(), int(is_dna), int(allow_ambiguity), int(ignore_softmask),
            t.ctypes.data, t.size, width, sa.ctypes.data, lcp.ctypes.data, sa.size,
            int(max_query_len is not None), int(max_query_len or 0),
            seed_mask.encode() if seed_mask is not None else None,
            starts.ctypes.data, starts.size, names, err, len(err))
        if rc != 0:
            raise RuntimeError(err.value.decode())

    def create(self, fasta, out, *, is_dna=False, allow_ambiguity=False, ignore_softmask=False,
               delimiter=b"%", seed_mask=None, max_query_len=None, num_partitions=16,
               random_seed=42, threads=1):
        """`sufr create` (sufr/src/lib.rs:321-371) end to end through the oracle."""
        text, starts, names = self.read_sequence_file(fasta, delimiter)
        norm = self.normalize(text, ignore_softmask)
        width = 4 if len(text) < 0xFFFFFFFF else 8          # suffix_array.rs:461
        sa, lcp, st = self.build(norm, is_dna=is_dna, allow_ambiguity=allow_ambiguity,
                                 max_query_len=max_query_len, seed_mask=seed_mask,
                                 num_partitions=num_partitions, random_seed=random_seed,
                                 threads=threads, width=width)
        self.write_file(out, is_dna=is_dna, allow_ambiguity=allow_ambiguity,
                        ignore_softmask=ignore_softmask, norm_text=norm, sa=sa, lcp=lcp,
                        width=width, max_query_len=max_query_len, seed_mask=seed_mask,
                        sequence_starts=starts, sequence_names=names)
        return st


# ----------------------------------------------------------------------
# independent witness: naive sort + exact LCP
# ----------------------------------------------------------------------
def eligible_positions(norm: np.ndarray, is_dna: bool, allow_ambiguity: bool) -> np.ndarray:
    """Eligibility predicate, sufr_builder.rs:446-449."""
    if not is_dna or allow_ambiguity:
        return np.arange(norm.size, dtype=np.int64)
    m = (norm == ord("$")) | (norm == ord("A")) | (norm == ord("C")) | (norm == ord("G")) | (norm == ord("T"))
    return np.nonzero(m)[0].astype(np.int64)


def naive_sa_lcp(norm: np.ndarray, is_dna=False, allow_ambiguity=False):
    """sorted(eligible, key=text[i:]) on raw bytes; LCP[0]=0, exact adjacent LCP.
    Pure Python: small inputs only (<= ~1e5 with short LCPs)."""
    t = norm.tobytes()
    pos = eligible_positions(norm, is_dna, allow_ambiguity).tolist()
    pos.sort(key=lambda i: t[i:])
    lcp = [0] * len(pos)
    n = len(t)
    for k in range(1, len(pos)):
        a, b = pos[k - 1], pos[k]
        m = min(n - a, n - b); c = 0
        while c < m and t[a + c] == t[b + c]:
            c += 1
        lcp[k] = c
    return np.asarray(pos, dtype=np.uint64), np.asarray(lcp, dtype=np.uint64)


def check_sa_lcp_properties(norm: np.ndarray, sa: np.ndarray, lcp: np.ndarray, *, is_dna, allow_ambiguity,
                            sample: int | None = None, seed: int = 0):
    """Size-independent checks used at BASELINE sizes: SA is a permutation of the
    eligible positions; for (sampled) adjacent ranks the LCP value is exact and
    the byte after it orders the pair."""
    n = norm.size
    elig = eligible_positions(norm, is_dna, allow_ambiguity)
    assert sa.size == elig.size, (sa.size, elig.size)
    assert lcp.size == sa.size
    srt = np.sort(sa.astype(np.int64))
    assert np.array_equal(srt, elig), "SA is not a permutation of the eligible positions"
    if sa.size == 0:
        return
    assert int(lcp[0]) == 0
    if sample is None or sample >= sa.size - 1:
        ranks = np.arange(1, sa.size)
    else:
        ranks = np.random.default_rng(seed).integers(1, sa.size, size=sample)
    pad = np.concatenate([norm, np.zeros(1, dtype=np.uint8)])
    a = sa[ranks - 1].astype(np.int64); b = sa[ranks].astype(np.int64); l = lcp[ranks].astype(np.int64)
    assert np.all(a + l <= n) and np.all(b + l <= n)
    # byte after the common prefix must order the pair (end of text sorts lowest)
    ca = np.where(a + l < n, pad[np.minimum(a + l, n)].astype(np.int64), -1)
    cb = np.where(b + l < n, pad[np.minimum(b + l, n)].astype(np.int64), -1)
    assert np.all(ca < cb), "adjacent suffixes out of order or LCP too small"
    # the common prefix itself must match: check in chunks, vectorised
    maxl = int(l.max()) if l.size else 0
    step = 0
    while step < maxl:
        live = np.nonzero(l > step)[0]
        if live.size == 0:
            break
        w = min(64, maxl - step)
        for k in range(w):
            lv = live[l[live] > step + k]
            if lv.size == 0:
                break
            assert np.array_equal(norm[a[lv] + step + k], norm[b[lv] + step + k]), "LCP too large"
        step += w


# ----------------------------------------------------------------------
# test-side .sufr v6 decoder
# ----------------------------------------------------------------------
@dataclass
class SufrFile:
    version: int
    is_dna: bool
    allow_ambiguity: bool
    ignore_softmask: bool
    text_len: int
    text_pos: int
    sa_pos: int
    lcp_pos: int
    num_suffixes: int
    max_query_len: int
    num_sequences: int
    sequence_starts: list
    seed_mask: bytes
    text: bytes
    sa: np.ndarray
    lcp: np.ndarray
    sequence_names: list = field(default_factory=list)
    width: int = 4


def parse_sufr(path) -> SufrFile:
    b = Path(path).read_bytes()
    ver, dna, amb, soft = b[0], b[1], b[2], b[3]
    text_len, text_pos, sa_pos, lcp_pos, nsuf, mql, nseq = struct.unpack_from("<7Q", b, 4)
    width = 4 if text_len < 0xFFFFFFFF else 8
    dt = np.dtype("<u4") if width == 4 else np.dtype("<u8")
    off = 60
    starts = np.frombuffer(b, dtype=dt, count=nseq, offset=off).tolist(); off += width * nseq
    (mlen,) = struct.unpack_from("<Q", b, off); off += 8
    mask = b[off:off + mlen]; off += mlen
    assert off == text_pos, (off, text_pos)
    text = b[text_pos:text_pos + text_len]
    assert text_pos + text_len == sa_pos
    sa = np.frombuffer(b, dtype=dt, count=nsuf, offset=sa_pos)
    assert sa_pos + width * nsuf == lcp_pos
    lcp = np.frombuffer(b, dtype=dt, count=nsuf, offset=lcp_pos)
    off = lcp_pos + width * nsuf
    (cnt,) = struct.unpack_from("<Q", b, off); off += 8
    names = []
    for _ in range(cnt):
        (ln,) = struct.unpack_from("<Q", b, off); off += 8
        names.append(b[off:off + ln].decode()); off += ln
    assert off == len(b), (off, len(b))
    return SufrFile(ver, bool(dna), bool(amb), bool(soft), text_len, text_pos, sa_pos, lcp_pos, nsuf,
                    mql, nseq, starts, mask, text, sa, lcp, names, width)


# generating commands of the golden files: reference mk_test_files.py:63-89 (all -n 16 -r 42)
GOLDEN_CASES = {
    "1.sufr": dict(fa="1.fa", is_dna=True),
    "2.sufr": dict(fa="2.fa", is_dna=True),
    "3.sufr": dict(fa="3.fa", is_dna=True),
    "2d.sufr": dict(fa="2.fa", is_dna=True, delimiter=b"N"),
    "abba.sufr": dict(fa="abba.fa"),
    "1n.sufr": dict(fa="1.fa", is_dna=True, allow_ambiguity=True),
    "2n.sufr": dict(fa="2.fa", is_dna=True, allow_ambiguity=True),
    "1s.sufr": dict(fa="1.fa", is_dna=True, ignore_softmask=True),  # current-format, unreferenced by the reference's tests
    "2s.sufr": dict(fa="2.fa", is_dna=True, ignore_softmask=True),
    "2ns.sufr": dict(fa="2.fa", is_dna=True, allow_ambiguity=True, ignore_softmask=True),
    "long_dna_sequence.sufr": dict(fa="long_dna_sequence.fa", is_dna=True),
    "long_dna_sequence_allow_ambiguity.sufr": dict(fa="long_dna_sequence.fa", is_dna=True, allow_ambiguity=True),
    "uniprot.sufr": dict(fa="uniprot.fa"),
    "uniprot-masked.sufr": dict(fa="uniprot.fa", seed_mask="10111011"),
}
