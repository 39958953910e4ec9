"""Prefix-bucket shards across GPUs: the boundary exchange and LCP stitch.

A genome sharded over N ranks (sufr_hip_sort_device_u32 with shard_index / num_shards) needs exactly one
exchange: every rank publishes {first suffix, last suffix, count} (24 bytes) so that rank r can compute
the LCP of its first suffix with the last suffix of the nearest non-empty shard before it -- the same
boundary fix SufrBuilder::write applies between partitions (sufr_builder.rs:893-902) -- and knows its
output offset (sum of the earlier counts).  `dist` is torch.distributed (backend nccl = RCCL on GPUs,
gloo in the CPU tests)."""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

Boundary = Tuple[int, int, int]   # (first suffix, last suffix, count)


def exchange_boundaries(first: int, last: int, count: int, device, dist=None) -> List[Boundary]:
    """all_gather of one (first, last, count) triple per rank; returns the list indexed by rank."""
    mine = torch.tensor([first, last, count], dtype=torch.int64, device=device)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [tuple(int(v) for v in mine.tolist())]
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [tuple(int(v) for v in t.tolist()) for t in out]


def gather_boundaries_device(sa, count: int, dist=None, always_collective: bool = False):
    """The same exchange without a host round trip: the triple is assembled on the device from the shard's own
    suffix array (sa: the int32 / int64 tensor sufr_hip_sort_device_* filled, `count` entries valid) and gathered into
    a [world, 3] int64 tensor that stays in device memory (backend nccl = RCCL; gloo accepts CPU tensors).
    always_collective: run the all_gather even in a process group of ONE rank (bench.py's forced-dist rehearsal: the
    RCCL call and its stream ordering against the stitch kernel execute on a one-GPU box)."""
    dev = sa.device
    if count:
        ends = torch.stack([sa[0], sa[count - 1]]).to(torch.int64) & 0xFFFFFFFF if sa.dtype == torch.int32 \
            else torch.stack([sa[0], sa[count - 1]]).to(torch.int64)
    else:
        ends = torch.zeros(2, dtype=torch.int64, device=dev)
    mine = torch.cat([ends, torch.tensor([count], dtype=torch.int64, device=dev)])
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not always_collective):
        return mine.view(1, 3)
    out = torch.empty(dist.get_world_size(), 3, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out.view(-1), mine)
    return out


def stitch_device(ctx, text_len: int, bounds, rank: int, lcp) -> None:
    """LCP[0] of this rank's device-resident shard := exact LCP with the last suffix of the nearest non-empty shard
    before it (sufr_hip_stitch_device_u32: a kernel on the context's text; `bounds` = gather_boundaries_device's
    tensor, on the GPU).  The gather ran on torch's stream, the kernel runs on the context's: wait for the former."""
    from . import _lib
    world = bounds.shape[0]
    if world == 1 or rank == 0 or lcp.numel() == 0:           # (an empty shard has no first LCP -- and no array to point at)
        return
    torch.cuda.current_stream(bounds.device).synchronize()
    fn = _lib.lib().sufr_hip_stitch_device_u64 if lcp.dtype == torch.int64 else _lib.lib().sufr_hip_stitch_device_u32
    ctx.check(fn(ctx.handle, text_len, bounds.data_ptr(), rank, world, lcp.data_ptr()))


def output_offset(boundaries: List[Boundary], rank: int) -> int:
    """Position of this shard's first entry in the concatenated SA / LCP arrays."""
    return sum(b[2] for b in boundaries[:rank])


def stitched_first_lcp(boundaries: List[Boundary], rank: int,
                       norm_slice: Callable[[int, int], np.ndarray], text_len: int) -> Optional[int]:
    """LCP of this shard's first suffix with the last suffix of the previous non-empty shard
    (None for the globally first suffix, whose LCP is 0 by definition, and for empty shards).
    norm_slice(start, length) returns normalised text bytes."""
    if boundaries[rank][2] == 0:
        return None
    prev = [r for r in range(rank) if boundaries[r][2] > 0]
    if not prev:
        return None
    a, b = boundaries[prev[-1]][1], boundaries[rank][0]
    k, step = 0, 64
    while True:
        la = min(step, text_len - a - k); lb = min(step, text_len - b - k)
        m = min(la, lb)
        if m <= 0:
            return k
        x = norm_slice(a + k, m); y = norm_slice(b + k, m)
        neq = np.nonzero(x != y)[0]
        if neq.size:
            return k + int(neq[0])
        k += m
        step = min(step * 4, 1 << 20)


def write_plan(boundaries: List[Boundary], rank: int):
    """Where rank `rank` writes in the one output file and what it stitches: (suffix offset of its slice, suffixes
    of the whole file, has_prev, last suffix of the nearest non-empty shard before it) -- the multi-writer form of
    the partition loop of SufrBuilder::write (sufr_builder.rs:875-906)."""
    total = sum(b[2] for b in boundaries)
    offset = output_offset(boundaries, rank)
    prev = [r for r in range(rank) if boundaries[r][2] > 0]
    has_prev = bool(prev) and boundaries[rank][2] > 0
    return offset, total, has_prev, (boundaries[prev[-1]][1] if has_prev else 0)


def create_sharded(ctx, seq, args, outfile: str, rank: int, world: int, dist, device):
    """`sufr create` with one process per GPU: this rank builds shard `rank` of `world` (C ABI:
    sufr_hip_shard_build), the ranks all_gather {first, last, count} (24 bytes each -- the only collective), rank 0
    lays out the file (sufr_write_frame), and after a barrier every rank streams its SA / LCP slice to its own range
    (sufr_hip_shard_write, which also sets the slice's first LCP to the boundary LCP).  seq: _lib.SequenceData, args:
    _lib.CreateArgs.  Returns (boundaries, stats)."""
    import ctypes as C
    import os
    from . import _lib
    L = _lib.lib()
    info = _lib.ShardInfo(); st = _lib.Stats()
    multi = dist is not None and dist.is_initialized() and world > 1
    path = os.fsencode(outfile)

    def agree(err):
        """Every rank learns whether ANY rank failed this step (one all_reduce): a rank that raised alone would leave
        the others waiting in the next collective for ever.  Rank 0 removes the output, then everybody raises."""
        flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device=device)
        if multi:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            if rank == 0:
                try:
                    os.unlink(outfile)
                except OSError:
                    pass
            if err is not None:
                raise err
            raise _lib.SufrHipError(-1, "another rank failed; the output file was removed")

    err = None
    try:
        ctx.check(L.sufr_hip_shard_build(ctx.handle, C.byref(seq), C.byref(args), rank, world, C.byref(info), C.byref(st)))
    except Exception as e:          # noqa: BLE001 -- reported to every rank below
        err = e
    agree(err)
    bounds = exchange_boundaries(int(info.first_suffix), int(info.last_suffix), int(info.num_suffixes), device, dist)
    offset, total, has_prev, prev_last = write_plan(bounds, rank)
    err = None
    if rank == 0:
        buf = C.create_string_buffer(512)
        rc = L.sufr_write_frame(path, C.byref(seq), C.byref(args), total, buf, len(buf))
        if rc != 0:
            err = _lib.SufrHipError(rc, buf.value.decode())
    agree(err)                      # (also the barrier: the frame exists before any rank writes its slice)
    err = None
    try:
        ctx.check(L.sufr_hip_shard_write(ctx.handle, C.byref(seq), C.byref(args), path, int(info.num_suffixes), total,
                                         offset, int(has_prev), prev_last, int(rank == 0)))
    except Exception as e:          # noqa: BLE001
        err = e
    agree(err)                      # a failed slice leaves a valid header over missing arrays: the file goes
    return bounds, st
