"""SuffixArray facade (libsufr/src/suffix_array.rs): `write` builds on the GPU, `read` opens a file for queries."""
from __future__ import annotations

from typing import Optional

from .sufr_builder import SufrBuilder, U32_MAX
from .sufr_file import SufrFile
from .types import SufrBuilderArgs


class SuffixArray:
    """After `read`, count / extract / list / locate / metadata / string_at are SufrFile's (suffix_array.rs:181-440)."""

    def __init__(self, inner: SufrFile):
        self.inner = inner

    @staticmethod
    def write(args: SufrBuilderArgs) -> str:
        """Build and write the .sufr file; u32 indices iff len(text) < u32::MAX.  Returns the path."""
        width = 4 if len(args.text) < U32_MAX else 8
        return SufrBuilder(args, index_width=width).path

    @staticmethod
    def read(filename: str, low_memory: bool = False) -> "SuffixArray":
        return SuffixArray(SufrFile(filename, low_memory))

    def count(self, queries, max_query_len: Optional[int] = None, low_memory: bool = False):
        return self.inner.count(queries, max_query_len, low_memory)

    def locate(self, queries, max_query_len: Optional[int] = None, low_memory: bool = False):
        return self.inner.locate(queries, max_query_len, low_memory)

    def extract(self, queries, max_query_len: Optional[int] = None, low_memory: bool = False, prefix_len=None, suffix_len=None):
        return self.inner.extract(queries, max_query_len, low_memory, prefix_len, suffix_len)

    def list(self, **opts):
        return self.inner.list(**opts)

    def metadata(self):
        return self.inner.metadata()

    def string_at(self, pos: int, len: Optional[int] = None) -> str:
        return self.inner.string_at(pos, len)
