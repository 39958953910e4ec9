"""Argument / result types of the construction path (reference: libsufr/src/types.rs)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

OUTFILE_VERSION = 6          # types.rs:16
SENTINEL_CHARACTER = b"$"    # types.rs:20


@dataclass
class SufrBuilderArgs:
    """libsufr::types::SufrBuilderArgs (types.rs:527-582): same names, same meaning."""
    text: bytes
    path: Optional[str] = None
    low_memory: bool = True
    max_query_len: Optional[int] = None
    is_dna: bool = False
    allow_ambiguity: bool = False
    ignore_softmask: bool = False
    sequence_starts: List[int] = field(default_factory=lambda: [0])
    sequence_names: List[str] = field(default_factory=lambda: ["1"])
    num_partitions: int = 16
    seed_mask: Optional[str] = None
    random_seed: int = 42


@dataclass
class SequenceFileData:
    """libsufr::types::SequenceFileData: result of util::read_sequence_file."""
    seq: bytes
    start_positions: List[int]
    sequence_names: List[str]
