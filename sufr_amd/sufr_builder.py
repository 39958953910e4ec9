"""SufrBuilder: the reference's build entry point (libsufr/src/sufr_builder.rs:38-220) on the MI355X path.

`SufrBuilder(args)` normalises the text, sorts the suffixes and computes the LCP array on the GPU and
writes the .sufr file, exactly like `SufrBuilder::<T>::new(args)`; the public fields carry the same
names.  `DeviceBuilder` is the device-resident form used by bench.py and the multi-GPU driver."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from .types import OUTFILE_VERSION, SufrBuilderArgs

U32_MAX = 0xFFFFFFFF


def _flags(is_dna: bool, allow_ambiguity: bool, ignore_softmask: bool, raw: bool) -> int:
    return ((_lib.FLAG_DNA if is_dna else 0) | (_lib.FLAG_ALLOW_AMBIGUITY if allow_ambiguity else 0) |
            (_lib.FLAG_IGNORE_SOFTMASK if ignore_softmask else 0) | (_lib.FLAG_RAW_TEXT if raw else 0))


class SufrBuilder:
    """Mirror of `SufrBuilder<T>`; `index_width` (4 or 8) plays the role of T."""

    def __init__(self, args: SufrBuilderArgs, index_width: Optional[int] = None, ctx: Optional[_lib.Context] = None,
                 write: bool = True):
        # SufrBuilder::new bails when both are Some(..) -- Some(0) included (sufr_builder.rs:163-165)
        if args.max_query_len is not None and args.seed_mask is not None:
            raise _lib.SufrHipError(-8, "Cannot use max_query_len and seed_mask together")
        L = _lib.lib()
        raw = np.frombuffer(args.text, dtype=np.uint8) if not isinstance(args.text, np.ndarray) else \
            np.ascontiguousarray(args.text, dtype=np.uint8)
        n = raw.size
        if index_width is None:
            index_width = 4 if n < U32_MAX else 8            # suffix_array.rs:461
        if index_width not in (4, 8):
            raise ValueError("index_width must be 4 or 8")
        own = ctx is None
        ctx = ctx or _lib.Context(0)
        try:
            self.version = OUTFILE_VERSION
            self.is_dna = args.is_dna
            self.allow_ambiguity = args.allow_ambiguity
            self.ignore_softmask = args.ignore_softmask
            self.text_len = n
            self.num_sequences = len(args.sequence_starts)
            self.sequence_starts = list(args.sequence_starts)
            self.sequence_names = list(args.sequence_names)
            self.max_query_len = args.max_query_len
            self.seed_mask = args.seed_mask
            self.path = args.path or "out.sufr"               # sufr_builder.rs:215
            self.index_width = index_width
            dt = np.uint32 if index_width == 4 else np.uint64
            norm = np.empty(n, dtype=np.uint8)
            sa = np.empty(n, dtype=dt)
            lcp = np.empty(n, dtype=dt)
            ns = C.c_uint64(0)
            self.stats = _lib.Stats()
            fn = L.sufr_hip_build_u32 if index_width == 4 else L.sufr_hip_build_u64
            ctx.check(fn(ctx.handle, raw.ctypes.data, n, _flags(args.is_dna, args.allow_ambiguity,
                                                               args.ignore_softmask, True),
                         int(args.max_query_len or 0) if args.max_query_len is not None else 0,
                         args.seed_mask.encode() if args.seed_mask is not None else None,
                         args.num_partitions, args.random_seed, norm.ctypes.data, sa.ctypes.data,
                         lcp.ctypes.data, n, C.byref(ns), C.byref(self.stats)))
            self.num_suffixes = ns.value
            self.text = norm
            self.suffix_array = sa[:ns.value]
            self.lcp = lcp[:ns.value]
            if write:
                self.write()
        finally:
            if own:
                ctx.close()

    def write(self) -> None:
        """SufrBuilder::write (sufr_builder.rs:817-918)."""
        L = _lib.lib()
        starts = np.asarray(self.sequence_starts, dtype=np.uint64)
        names = (C.c_char_p * len(self.sequence_names))(*[s.encode() for s in self.sequence_names])
        err = C.create_string_buffer(512)
        sa = np.ascontiguousarray(self.suffix_array)
        lcp = np.ascontiguousarray(self.lcp)
        rc = L.sufr_write_file(self.path.encode(), int(self.is_dna), int(self.allow_ambiguity),
                               int(self.ignore_softmask), self.text.ctypes.data, self.text_len, self.index_width,
                               sa.ctypes.data, lcp.ctypes.data, self.num_suffixes,
                               int(self.max_query_len is not None), int(self.max_query_len or 0),
                               self.seed_mask.encode() if self.seed_mask is not None else None,
                               starts.ctypes.data, starts.size, names, err, len(err))
        if rc != 0:
            raise _lib.SufrHipError(rc, err.value.decode())


class DeviceBuilder:
    """Device-resident build: text, SA and LCP stay in HBM (torch tensors are only the allocation)."""

    def __init__(self, device: int = 0):
        self.ctx = _lib.Context(device)
        self.device = device
        self.stats = _lib.Stats()

    def sort(self, d_text, *, is_dna=False, allow_ambiguity=False, ignore_softmask=False, raw_text=False,
             shard_index: int = 0, num_shards: int = 1, out_sa=None, out_lcp=None, num_partitions=16,
             random_seed=42, max_query_len=None, seed_mask=None, index_width: int = 4):
        """d_text: uint8 torch tensor on this GPU.  Returns (sa, lcp) int32-typed torch tensors holding
        u32 values (views of length num_suffixes); index_width=8: int64 tensors through sufr_hip_sort_device_u64
        (texts of 2^32 - 1 bytes and more need it, suffix_array.rs:461)."""
        import torch
        assert d_text.is_cuda and d_text.dtype == torch.uint8 and d_text.is_contiguous()
        # the context's stream is not ordered against torch's: whatever produced d_text must have finished
        torch.cuda.current_stream(d_text.device).synchronize()
        n = d_text.numel()
        cap = n if out_sa is None else out_sa.numel()
        if out_sa is None:
            dt = torch.int32 if index_width == 4 else torch.int64
            out_sa = torch.empty(n, dtype=dt, device=d_text.device)
            out_lcp = torch.empty(n, dtype=dt, device=d_text.device)
        ns = C.c_uint64(0)
        fn = _lib.lib().sufr_hip_sort_device_u32 if index_width == 4 else _lib.lib().sufr_hip_sort_device_u64
        rc = fn(
            self.ctx.handle, d_text.data_ptr(), n, _flags(is_dna, allow_ambiguity, ignore_softmask, raw_text),
            int(max_query_len or 0), seed_mask.encode() if seed_mask is not None else None, num_partitions,
            random_seed, shard_index, num_shards, out_sa.data_ptr(), out_lcp.data_ptr(),
            cap, C.byref(ns), C.byref(self.stats))
        self.ctx.check(rc)
        self.num_suffixes = ns.value
        return out_sa[:ns.value], out_lcp[:ns.value]

    def close(self):
        self.ctx.close()
