"""Reading and searching a .sufr file: the host mirror of libsufr's query side.

`SufrFile` follows SufrFile<T> / SuffixArray of the reference (libsufr/src/sufr_file.rs, suffix_array.rs:181-440):
count / locate / extract / list / metadata / string_at with the reference's option and result names.  The work is done
by the C ABI of include/sufr_query.h (a mapped file, two binary searches per query).  `DeviceIndex` answers batches of
queries on the GPU from text + suffix array resident in HBM (sufr_hip_search_batch)."""
from __future__ import annotations

import builtins
import ctypes as C
import datetime
from dataclasses import dataclass, field
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from ._lib import Context, FileMeta, SufrHipError, lib


@dataclass
class CountResult:                     # types.rs:365-374
    query_num: int
    query: str
    count: int


@dataclass
class LocatePosition:                  # types.rs:510-522
    suffix: int
    rank: int
    sequence_name: str
    sequence_position: int


@dataclass
class LocateResult:                    # types.rs:494-504
    query_num: int
    query: str
    positions: List[LocatePosition] = field(default_factory=list)


@dataclass
class ExtractSequence:                 # types.rs:423-444
    suffix: int
    rank: int
    sequence_name: str
    sequence_start: int
    sequence_range: Tuple[int, int]
    suffix_offset: int


@dataclass
class ExtractResult:                   # types.rs:400-409
    query_num: int
    query: str
    sequences: List[ExtractSequence] = field(default_factory=list)


@dataclass
class SufrMetadata:                    # types.rs:587-626
    filename: str
    modified: datetime.datetime
    file_size: int
    file_version: int
    is_dna: bool
    allow_ambiguity: bool
    ignore_softmask: bool
    text_len: int
    len_suffixes: int
    num_sequences: int
    sequence_starts: List[int]
    sequence_names: List[str]
    max_query_len: int                 # sort_type: MaxQueryLen(n) ...
    seed_mask: Optional[str]           # ... or Mask(seed mask)


def _as_bytes(q) -> bytes:
    return q.encode() if isinstance(q, str) else bytes(q)


class SufrFile:
    """An open version-6 .sufr file (SufrFile::read, sufr_file.rs:145-275).  The file is mapped, `low_memory` /
    `very_low_memory` of the reference only choose how much of it the reference copies to memory and are accepted
    and ignored here."""

    def __init__(self, filename: str, low_memory: bool = False):
        L = lib()
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = L.sufr_file_open(str(filename).encode(), C.byref(h), err, len(err))
        if rc != 0:
            raise SufrHipError(rc, err.value.decode())
        import threading
        self._view_lock = threading.RLock()       # re-entrant: a GC pass inside _view() can finalise a dead view of this file on the same thread
        self._live_views = 0
        self._close_pending = False
        self._h = h
        self.filename = str(filename)
        m = FileMeta()
        L.sufr_file_metadata(h, C.byref(m))
        self._meta = m
        self.text_len, self.len_suffixes, self.num_sequences = m.text_len, m.len_suffixes, m.num_sequences
        self.index_width = m.index_width
        self.is_dna, self.allow_ambiguity, self.ignore_softmask = bool(m.is_dna), bool(m.allow_ambiguity), bool(m.ignore_softmask)
        self.max_query_len = m.max_query_len
        self.sequence_starts = [L.sufr_file_sequence_start(h, i) for i in range(m.num_sequences)]
        self.sequence_names = [L.sufr_file_sequence_name(h, i).decode() for i in range(m.num_sequences)]
        self.seed_mask = None
        if m.seed_mask_len:
            raw = C.string_at(L.sufr_file_seed_mask(h), m.seed_mask_len)
            self.seed_mask = "".join("1" if b == 1 else "0" for b in raw)

    # -- views into the mapping --------------------------------------------------------------------------------------
    # Zero-copy: the arrays alias the mapped file, and every view PINS the mapping -- it holds a reference to this
    # object (so `SufrFile(p).suffix_array.tolist()` works).  close() (and leaving a `with` block) is therefore a
    # REQUEST while views are alive: the file stays mapped and open until the last view is gone (`pending_close` is
    # True meanwhile, `closed` only afterwards).  A caller that must release the file at a known point -- before
    # re-creating or truncating the same path, which would turn reads of a stale view into SIGBUS -- drops its views
    # first or takes copies (`copy=True` of array()).  The view count is guarded by a lock: finalisers run on
    # whichever thread drops the last reference.
    def _view(self, getter, count, dtype):
        import weakref
        with self._view_lock:
            if self._h is None:
                raise ValueError("SufrFile is closed")
            if count == 0:
                return np.empty(0, dtype=dtype)
            buf = (C.c_uint8 * (count * np.dtype(dtype).itemsize)).from_address(getter(self._h))
            self._live_views += 1
        buf._owner = self
        arr = np.frombuffer(buf, dtype=dtype)
        weakref.finalize(buf, SufrFile._view_gone, self)
        return arr

    @staticmethod
    def _view_gone(owner):
        with owner._view_lock:
            owner._live_views -= 1
            last = owner._live_views == 0 and owner._close_pending
        if last:
            owner.close()

    @property
    def closed(self) -> bool:
        """The mapping is gone (no view outstanding, close() done)."""
        return self._h is None

    @property
    def pending_close(self) -> bool:
        """close() was called while views were alive: the file is unmapped when the last of them goes."""
        return self._close_pending and self._h is not None

    def array(self, which: str, copy: bool = False) -> np.ndarray:
        """'text' | 'suffix_array' | 'lcp'; copy=True returns an array of its own (it does not pin the mapping)."""
        a = {"text": lambda: self.text, "suffix_array": lambda: self.suffix_array, "lcp": lambda: self.lcp}[which]()
        return a.copy() if copy else a

    @property
    def text(self) -> np.ndarray:
        return self._view(lib().sufr_file_text, self.text_len, np.uint8)

    @property
    def suffix_array(self) -> np.ndarray:
        return self._view(lib().sufr_file_suffix_array, self.len_suffixes, np.uint32 if self.index_width == 4 else np.uint64)

    @property
    def lcp(self) -> np.ndarray:
        return self._view(lib().sufr_file_lcp_array, self.len_suffixes, np.uint32 if self.index_width == 4 else np.uint64)

    def close(self):
        lock = getattr(self, "_view_lock", None)
        if lock is None:                                  # (__init__ failed before the mapping existed)
            return
        with lock:
            if self._live_views > 0:                      # arrays still alias the mapping: unmap when the last one goes
                self._close_pending = True
                return
            h, self._h = self._h, None
            self._close_pending = False
        if h:
            lib().sufr_file_close(h)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the query API -------------------------------------------------------------------------------------------------
    def search(self, query, max_query_len: Optional[int] = None) -> Optional[Tuple[int, int]]:
        """Half-open rank range of the suffixes that match `query`, or None (SufrSearch::search, sufr_search.rs:104-168)."""
        q = _as_bytes(query)
        lo, hi = C.c_uint64(), C.c_uint64()
        hit = lib().sufr_file_search(self._h, q, len(q), int(max_query_len is not None), max_query_len or 0,
                                     C.byref(lo), C.byref(hi))
        return (lo.value, hi.value) if hit else None

    def search_batch(self, queries: Sequence, max_query_len: Optional[int] = None, threads: int = 0):
        """rank_lo, rank_hi (uint64 arrays, lo == hi == 0: not found) for a batch, `threads` host workers (0: one per core):
        the rayon loop of SufrFile::count / locate (sufr_file.rs:760-800)."""
        qb, off = pack_queries(queries)
        lo = np.zeros(len(off) - 1, dtype=np.uint64)
        hi = np.zeros(len(off) - 1, dtype=np.uint64)
        rc = lib().sufr_file_search_batch(self._h, qb.ctypes.data, off.ctypes.data, len(off) - 1, int(max_query_len is not None),
                                          max_query_len or 0, lo.ctypes.data, hi.ctypes.data, threads)
        if rc != 0:
            raise SufrHipError(rc, "sufr_file_search_batch failed")
        return lo, hi

    def count(self, queries: Sequence, max_query_len: Optional[int] = None, low_memory: bool = False) -> List[CountResult]:
        lo, hi = self.search_batch(queries, max_query_len)
        return [CountResult(i, q if isinstance(q, str) else bytes(q).decode("latin-1"), int(hi[i] - lo[i])) for i, q in enumerate(queries)]

    def _sequence_of(self, suffix: int) -> int:
        return lib().sufr_file_sequence_of(self._h, suffix)

    def locate(self, queries: Sequence, max_query_len: Optional[int] = None, low_memory: bool = False) -> List[LocateResult]:
        """SufrFile::locate (sufr_file.rs:1110-1175): positions in rank order."""
        sa = self.suffix_array
        out = []
        for i, q in enumerate(queries):
            res = LocateResult(i, q if isinstance(q, str) else bytes(q).decode("latin-1"))
            r = self.search(q, max_query_len)
            if r:
                for rank in range(r[0], r[1]):
                    sfx = int(sa[rank])
                    k = self._sequence_of(sfx)
                    res.positions.append(LocatePosition(sfx, rank, self.sequence_names[k], sfx - self.sequence_starts[k]))
            out.append(res)
        return out

    def extract(self, queries: Sequence, max_query_len: Optional[int] = None, low_memory: bool = False,
                prefix_len: Optional[int] = None, suffix_len: Optional[int] = None) -> List[ExtractResult]:
        """SufrFile::extract (sufr_file.rs:898-960)."""
        sa = self.suffix_array
        out = []
        for i, q in enumerate(queries):
            res = ExtractResult(i, q if isinstance(q, str) else bytes(q).decode("latin-1"))
            r = self.search(q, max_query_len)
            if r:
                for rank in range(r[0], r[1]):
                    sfx = int(sa[rank])
                    k = self._sequence_of(sfx)
                    start = self.sequence_starts[k]
                    end = self.sequence_starts[k + 1] if k + 1 < self.num_sequences else self.text_len
                    rel = sfx - start
                    cstart = max(rel - (prefix_len or 0), 0)
                    cend = min(rel + suffix_len, end) if suffix_len is not None else end
                    res.sequences.append(ExtractSequence(sfx, rank, self.sequence_names[k], start, (cstart, cend), rel - cstart))
            out.append(res)
        return out

    def string_at(self, pos: int, length: Optional[int] = None) -> str:
        """SufrFile::string_at (sufr_file.rs:399-411)."""
        end = min(pos + length, self.text_len) if length is not None else self.text_len
        return bytes(self.text[pos:end]).decode("latin-1")

    def list(self, ranks: Iterable[int] = (), show_rank=False, show_suffix=False, show_lcp=False, len: Optional[int] = None,
             number: Optional[int] = None) -> List[str]:
        """The lines `sufr list` prints (SufrFile::list, sufr_file.rs:1013-1077)."""
        width = builtins.len(str(self.text_len))
        sa, lcp = self.suffix_array, self.lcp
        n = self.text_len if len is None else len
        ranks = list(ranks)
        if not ranks:
            ranks = range(self.len_suffixes if not number else min(number, self.len_suffixes))
        lines = []
        for r in ranks:
            if r >= self.len_suffixes:
                continue
            sfx = int(sa[r])
            cols = []
            if show_rank:
                cols.append(f"{r:>{width}} ")
            if show_suffix:
                cols.append(f"{sfx:>{width}} ")
            if show_lcp:
                cols.append(f"{int(lcp[r]):>{width}} ")
            lines.append("".join(cols) + self.string_at(sfx, n))
        return lines

    def metadata(self) -> SufrMetadata:
        m = self._meta
        return SufrMetadata(self.filename, datetime.datetime.fromtimestamp(m.modified), m.file_size, m.version, self.is_dna,
                            self.allow_ambiguity, self.ignore_softmask, m.text_len, m.len_suffixes, m.num_sequences,
                            list(self.sequence_starts), list(self.sequence_names), m.max_query_len, self.seed_mask)



def pack_queries(queries: Sequence) -> Tuple[np.ndarray, np.ndarray]:
    """Concatenated query bytes + offsets, the batch layout of sufr_hip_search_batch."""
    bs = [_as_bytes(q) for q in queries]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    return np.frombuffer(b"".join(bs), dtype=np.uint8).copy(), off


class DeviceIndex:
    """Text + suffix array resident in HBM, searched a batch at a time (include/sufr_query.h, device section).

    DeviceIndex.load(ctx, sufr_file)                              copies an open file to the GPU
    DeviceIndex.wrap(ctx, text_tensor, sa_tensor, ...)            wraps torch CUDA tensors (e.g. DeviceBuilder output)"""

    def __init__(self, ctx: Context, handle, keep=()):
        self.ctx, self._h, self._keep = ctx, handle, keep

    @classmethod
    def load(cls, ctx: Context, f: SufrFile) -> "DeviceIndex":
        h = C.c_void_p()
        ctx.check(lib().sufr_hip_index_load(ctx.handle, f._h, C.byref(h)))
        ix = cls(ctx, h)
        ix.index_width = lib().sufr_hip_index_width(h)     # (= f.index_width: the file format's rule)
        return ix

    @classmethod
    def wrap(cls, ctx: Context, text, sa, max_query_len: int = 0, seed_mask: Optional[str] = None, is_dna: bool = False,
             prefix_table: bool = True) -> "DeviceIndex":
        import torch
        if not (text.is_cuda and sa.is_cuda and text.dtype == torch.uint8 and
                sa.dtype in (torch.int32, torch.uint32, torch.int64, torch.uint64)):
            raise ValueError("wrap() takes a uint8 text and a 32- or 64-bit suffix array on the GPU")
        wide = sa.dtype in (torch.int64, torch.uint64)
        if (text.numel() >= 0xFFFFFFFF) and not wide:
            raise ValueError("texts of 2^32 - 1 bytes and more have 64-bit suffix arrays (suffix_array.rs:460-470)")
        h = C.c_void_p()
        from ._lib import FLAG_DNA, FLAG_NO_PREFIX_TABLE, FLAG_SA_U64
        torch.cuda.current_stream(text.device).synchronize()      # the table is built from the arrays right away
        flags = (FLAG_DNA if is_dna else 0) | (0 if prefix_table else FLAG_NO_PREFIX_TABLE) | (FLAG_SA_U64 if wide else 0)
        ctx.check(lib().sufr_hip_index_wrap(ctx.handle, text.data_ptr(), text.numel(), sa.data_ptr(), sa.numel(), flags,
                                            max_query_len, seed_mask.encode() if seed_mask else None, C.byref(h)))
        ix = cls(ctx, h, keep=(text, sa))
        ix.index_width = lib().sufr_hip_index_width(h)     # 8 iff the array was taken as 64-bit
        assert ix.index_width == (8 if wide else 4)
        return ix

    def close(self):
        if getattr(self, "_h", None):
            lib().sufr_hip_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search(self, queries: Sequence, max_query_len: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
        """rank_lo, rank_hi (uint64 arrays; lo == hi == 0 where the query does not occur)."""
        qb, off = pack_queries(queries)
        return self.search_packed(qb, off, max_query_len)

    def search_packed(self, qbytes: np.ndarray, offsets: np.ndarray, max_query_len: Optional[int] = None):
        nq = len(offsets) - 1
        lo = np.zeros(nq, dtype=np.uint64)
        hi = np.zeros(nq, dtype=np.uint64)
        qbytes = np.ascontiguousarray(qbytes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.ctx.check(lib().sufr_hip_search_batch(self.ctx.handle, self._h, qbytes.ctypes.data, offsets.ctypes.data, nq,
                                                  int(max_query_len is not None), max_query_len or 0,
                                                  lo.ctypes.data, hi.ctypes.data))
        return lo, hi

    def search_device(self, qbytes, offsets, max_query_len: Optional[int] = None, wait: bool = True):
        """torch CUDA tensors in (uint8 bytes, int64 offsets), torch CUDA tensors out.  The launch goes to the context's
        stream, which is not ordered against torch's: the producer of the inputs is synchronised first, and with `wait`
        the answers are complete on return (wait=False: call ctx.synchronize() before reading them)."""
        import torch
        torch.cuda.current_stream(qbytes.device).synchronize()
        nq = offsets.numel() - 1
        lo = torch.empty(nq, dtype=torch.int64, device=qbytes.device)
        hi = torch.empty(nq, dtype=torch.int64, device=qbytes.device)
        self.ctx.check(lib().sufr_hip_search_batch_device(self.ctx.handle, self._h, qbytes.data_ptr(), offsets.data_ptr(), nq,
                                                         int(max_query_len is not None), max_query_len or 0,
                                                         lo.data_ptr(), hi.data_ptr()))
        if wait:
            self.ctx.synchronize()
        return lo, hi

    def locate_device(self, lo, hi, max_hits: int = 0, capacity: Optional[int] = None):
        """Positions behind rank ranges that are on the device (torch int64 tensors from search_device): returns
        (offsets int64[nq + 1], positions int32 holding u32 values); query i owns positions[offsets[i]:offsets[i + 1]],
        in rank order, at most max_hits of them (0: all)."""
        import torch
        nq = lo.numel()
        off = torch.empty(nq + 1, dtype=torch.int64, device=lo.device)
        total = C.c_uint64(0)
        torch.cuda.current_stream(lo.device).synchronize()
        if capacity is None:                                   # size the output from the counts
            cnt = hi - lo
            capacity = int((cnt.clamp(max=max_hits) if max_hits else cnt).sum())
        pos = torch.empty(max(capacity, 1), dtype=torch.int64 if getattr(self, "index_width", 4) == 8 else torch.int32,
                          device=lo.device)
        self.ctx.check(lib().sufr_hip_locate_batch_device(self.ctx.handle, self._h, lo.data_ptr(), hi.data_ptr(), nq, max_hits,
                                                         off.data_ptr(), pos.data_ptr(), capacity, C.byref(total)))
        self.ctx.synchronize()
        return off, pos[:total.value]

    def locate(self, queries: Sequence, max_query_len: Optional[int] = None, max_hits: int = 0) -> List[np.ndarray]:
        """Text positions of every query's matches in rank order (uint32 arrays), searched and gathered on the device."""
        import torch
        qb, off = pack_queries(queries)
        dev = torch.device("cuda", self.ctx.device)
        lo, hi = self.search_device(torch.from_numpy(qb).to(dev), torch.from_numpy(off.astype(np.int64)).to(dev), max_query_len)
        o, p = self.locate_device(lo, hi, max_hits)
        o = o.cpu().numpy(); p = p.cpu().numpy().view(np.uint32)
        return [p[o[i]:o[i + 1]] for i in range(len(queries))]

    def count(self, queries: Sequence, max_query_len: Optional[int] = None) -> List[CountResult]:
        lo, hi = self.search(queries, max_query_len)
        return [CountResult(i, q if isinstance(q, str) else bytes(q).decode("latin-1"), int(hi[i] - lo[i]))
                for i, q in enumerate(queries)]
