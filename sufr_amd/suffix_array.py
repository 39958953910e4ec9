"""SuffixArray facade: only the construction entry (libsufr/src/suffix_array.rs:460-470)."""
from __future__ import annotations

from .sufr_builder import SufrBuilder, U32_MAX
from .types import SufrBuilderArgs


class SuffixArray:
    @staticmethod
    def write(args: SufrBuilderArgs) -> str:
        """Build and write the .sufr file; u32 indices iff len(text) < u32::MAX.  Returns the path."""
        width = 4 if len(args.text) < U32_MAX else 8
        return SufrBuilder(args, index_width=width).path
