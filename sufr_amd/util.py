"""libsufr::util equivalents that sit on the construction path."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .types import SequenceFileData


def read_sequence_file(path: str | os.PathLike, sequence_delimiter: bytes | int = b"%") -> SequenceFileData:
    """util::read_sequence_file (util.rs:51-89): FASTA/FASTQ -> text + '$', starts, names."""
    L = _lib.lib()
    d = sequence_delimiter if isinstance(sequence_delimiter, int) else sequence_delimiter[0]
    sd = _lib.SequenceData()
    err = C.create_string_buffer(512)
    rc = L.sufr_read_sequence_file(os.fsencode(path), d, C.byref(sd), err, len(err))
    if rc != 0:
        raise _lib.SufrHipError(rc, err.value.decode())
    try:
        seq = C.string_at(sd.seq, sd.seq_len)
        starts = [sd.start_positions[i] for i in range(sd.num_sequences)]
        names = [sd.sequence_names[i].decode() for i in range(sd.num_sequences)]
    finally:
        L.sufr_sequence_data_free(C.byref(sd))
    return SequenceFileData(seq, starts, names)


def normalize(text: bytes | np.ndarray, ignore_softmask: bool) -> np.ndarray:
    """Text map of SufrBuilder::new (sufr_builder.rs:144-160)."""
    a = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text)
    out = np.empty_like(a)
    rc = _lib.lib().sufr_hip_normalize(a.ctypes.data, out.ctypes.data, a.size, int(ignore_softmask))
    if rc != 0:
        raise _lib.SufrHipError(rc, "normalize failed")
    return out


def lcp_pair(norm_text: np.ndarray, a: int, b: int) -> int:
    """find_lcp(a, b, text_len, 0) of the boundary stitch (sufr_builder.rs:893-902)."""
    t = np.ascontiguousarray(norm_text, dtype=np.uint8)
    return int(_lib.lib().sufr_hip_lcp_pair(t.ctypes.data, t.size, a, b))
