"""`sufr create` (sufr/src/lib.rs:321-371) through the C ABI: read the sequence file, build on the GPU,
write the file.  The native CLI binary is csrc/_build/sufr; this is the same call from Python."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import _lib


def create(input: str, output: Optional[str] = None, *, num_partitions: int = 16,
           max_query_len: Optional[int] = None, is_dna: bool = False, allow_ambiguity: bool = False,
           ignore_softmask: bool = False, sequence_delimiter: str = "%", seed_mask: Optional[str] = None,
           random_seed: int = 42, device: int = 0):
    ctx = _lib.Context(device)
    try:
        a = _lib.CreateArgs(os.fsencode(input), os.fsencode(output) if output else None, num_partitions,
                            int(max_query_len is not None), int(max_query_len or 0), int(is_dna),
                            int(allow_ambiguity), int(ignore_softmask), ord(sequence_delimiter),
                            seed_mask.encode() if seed_mask else None, random_seed)
        path = C.create_string_buffer(4096)
        st = _lib.Stats()
        ctx.check(_lib.lib().sufr_hip_create_file(ctx.handle, C.byref(a), path, len(path), C.byref(st)))
        return path.value.decode(), st
    finally:
        ctx.close()
