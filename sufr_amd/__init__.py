"""sufr_amd: MI355X-native suffix-array + LCP construction behind the `sufr create` contract.

The construction path of TravisWheelerLab/sufr and, next to it, the reader / query side of
the files it writes (row f3); the compute is hand-written HIP for gfx950 in csrc/, reached through the C ABI of
include/sufr_hip.h and include/sufr_query.h."""
from ._lib import Context, Stats, SufrHipError, build_extension, lib, EXPORTS, QUERY_EXPORTS, LIB_PATH, CLI_PATH
from .types import OUTFILE_VERSION, SENTINEL_CHARACTER, SequenceFileData, SufrBuilderArgs
from .util import lcp_pair, normalize, read_sequence_file
from .sufr_builder import DeviceBuilder, SufrBuilder
from .sufr_file import (CountResult, DeviceIndex, ExtractResult, ExtractSequence, LocatePosition, LocateResult, SufrFile,
                        SufrMetadata, pack_queries)
from .suffix_array import SuffixArray
from .cli import create

__all__ = [
    "Context", "Stats", "SufrHipError", "build_extension", "lib", "EXPORTS", "LIB_PATH", "CLI_PATH",
    "OUTFILE_VERSION", "SENTINEL_CHARACTER", "SequenceFileData", "SufrBuilderArgs", "lcp_pair", "normalize",
    "read_sequence_file", "DeviceBuilder", "SufrBuilder", "SuffixArray", "create", "QUERY_EXPORTS", "SufrFile", "DeviceIndex",
    "CountResult", "LocateResult", "LocatePosition", "ExtractResult", "ExtractSequence", "SufrMetadata", "pack_queries",
]
