"""`sufr create` (sufr/src/lib.rs:321-371) through the C ABI: read the sequence file, build on the GPU,
write the file.  The native CLI binary is csrc/_build/sufr; this is the same call from Python."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

from . import _lib


def create(input: str, output: Optional[str] = None, *, num_partitions: int = 16,
           max_query_len: Optional[int] = None, is_dna: bool = False, allow_ambiguity: bool = False,
           ignore_softmask: bool = False, sequence_delimiter: str = "%", seed_mask: Optional[str] = None,
           random_seed: int = 42, device: int = 0, devices: Optional[Sequence[int]] = None):
    """devices: build on several GPUs (`sufr --devices 0,1,... create`): shard r of len(devices) on devices[r], every
    shard written into its own range of the one output file; an ordinal may repeat (several shards on one GPU)."""
    a = create_args(input, output, num_partitions=num_partitions, max_query_len=max_query_len, is_dna=is_dna,
                    allow_ambiguity=allow_ambiguity, ignore_softmask=ignore_softmask,
                    sequence_delimiter=sequence_delimiter, seed_mask=seed_mask, random_seed=random_seed)
    path = C.create_string_buffer(4096)
    if devices is None:
        ctx = _lib.Context(device)
        try:
            st = _lib.Stats()
            ctx.check(_lib.lib().sufr_hip_create_file(ctx.handle, C.byref(a), path, len(path), C.byref(st)))
            return path.value.decode(), st
        finally:
            ctx.close()
    ctxs = [_lib.Context(d) for d in devices]
    try:
        handles = (C.c_void_p * len(ctxs))(*[c.handle for c in ctxs])
        sts = (_lib.Stats * len(ctxs))()
        ctxs[0].check(_lib.lib().sufr_hip_create_file_multi(handles, len(ctxs), C.byref(a), path, len(path), sts))
        return path.value.decode(), list(sts)
    finally:
        for c in ctxs:
            c.close()


def create_args(input: str, output: Optional[str] = None, *, num_partitions: int = 16,
                max_query_len: Optional[int] = None, is_dna: bool = False, allow_ambiguity: bool = False,
                ignore_softmask: bool = False, sequence_delimiter: str = "%", seed_mask: Optional[str] = None,
                random_seed: int = 42) -> "_lib.CreateArgs":
    """CreateArgs (sufr/src/lib.rs:83-125) as the C struct"""
    return _lib.CreateArgs(os.fsencode(input), os.fsencode(output) if output else None, num_partitions,
                           int(max_query_len is not None), int(max_query_len or 0), int(is_dna),
                           int(allow_ambiguity), int(ignore_softmask), ord(sequence_delimiter),
                           seed_mask.encode() if seed_mask else None, random_seed)
