"""ctypes binding of libsufr_hip.so (include/sufr_hip.h).  No CPU fallback: if the HIP library is
missing or no GPU is visible, every build entry point raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = CSRC / "_build" / "libsufr_hip.so"
# profiles/*.sh may point the binding at the probes build (tuning knobs from SUFR_PROBE_*, phase stamps); the tests and
# the driver use the plain library, which reads no environment besides SUFR_HIP_DEBUG and SUFR_SERIAL_READER.  The
# switch is loud: a stray variable must not change a user's kernels silently.
if os.environ.get("SUFR_AMD_PROBES_LIB"):
    import sys as _sys
    LIB_PATH = CSRC / "_build" / "libsufr_hip_probes.so"
    print(f"sufr_amd: SUFR_AMD_PROBES_LIB is set -- loading the PROBES build {LIB_PATH.name} (tuning knobs from "
          "SUFR_PROBE_*; for profiling only)", file=_sys.stderr)
CLI_PATH = CSRC / "_build" / "sufr"
# tests/test_sanitized_host.py runs the host-side tests in a child process against the HOST-ONLY AddressSanitizer + UBSan
# build (`make asan`: reader, writer, query code; every device entry point answers "no device").  Loud, like the probes switch.
HOST_ASAN = bool(os.environ.get("SUFR_AMD_HOST_ASAN_LIB"))
if HOST_ASAN:
    import sys as _sys
    LIB_PATH = CSRC / "_build" / "libsufr_host_asan.so"
    CLI_PATH = CSRC / "_build" / "sufr_host_asan"
    print(f"sufr_amd: SUFR_AMD_HOST_ASAN_LIB is set -- loading the host-only sanitizer build {LIB_PATH.name} (no device code)",
          file=_sys.stderr)


class SufrHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{message} (sufr_hip error {code})")
        self.code = code
        self.message = message


class Stats(C.Structure):
    """sufr_hip_stats"""
    _fields_ = [
        ("text_len", C.c_uint64), ("num_suffixes", C.c_uint64),
        ("alphabet_size", C.c_uint32), ("bits_per_char", C.c_uint32), ("chars_per_key", C.c_uint32),
        ("digit_bits", C.c_uint32), ("num_passes", C.c_uint32), ("num_levels", C.c_uint32),
        ("num_large_groups", C.c_uint64), ("deep_records", C.c_uint64),
        ("top_lo", C.c_uint32), ("top_hi", C.c_uint32), ("partition_workgroups", C.c_uint32),
        ("partition_variant", C.c_uint32),
        ("ms_total", C.c_float), ("ms_normalize", C.c_float), ("ms_hist_text", C.c_float),
        ("ms_partition", C.c_float), ("ms_passes", C.c_float), ("ms_finish", C.c_float),
        ("ms_deep", C.c_float), ("host_read_s", C.c_float), ("host_build_s", C.c_float),
        ("host_write_s", C.c_float),
        ("num_exceptions", C.c_uint32), ("ms_exceptions", C.c_float), ("num_reinserted", C.c_uint64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class SequenceData(C.Structure):
    _fields_ = [("seq", C.POINTER(C.c_uint8)), ("seq_len", C.c_uint64),
                ("start_positions", C.POINTER(C.c_uint64)), ("sequence_names", C.POINTER(C.c_char_p)),
                ("num_sequences", C.c_uint64)]


class CreateArgs(C.Structure):
    _fields_ = [("input", C.c_char_p), ("output", C.c_char_p), ("num_partitions", C.c_uint64),
                ("has_max_query_len", C.c_int), ("max_query_len", C.c_uint64), ("is_dna", C.c_int),
                ("allow_ambiguity", C.c_int), ("ignore_softmask", C.c_int), ("sequence_delimiter", C.c_uint8),
                ("seed_mask", C.c_char_p), ("random_seed", C.c_uint64)]


class ShardInfo(C.Structure):
    _fields_ = [("num_suffixes", C.c_uint64), ("first_suffix", C.c_uint64), ("last_suffix", C.c_uint64)]


FLAG_DNA, FLAG_ALLOW_AMBIGUITY, FLAG_IGNORE_SOFTMASK, FLAG_RAW_TEXT = 1, 2, 4, 8
FLAG_NO_PREFIX_TABLE = 0x100      # sufr_hip_index_wrap only
FLAG_SA_U64 = 0x200               # sufr_hip_index_wrap only: 64-bit suffix array of a text below 2^32 - 1 bytes

# every symbol include/sufr_hip.h declares
EXPORTS = [
    "sufr_hip_abi_version", "sufr_hip_device_count", "sufr_hip_create", "sufr_hip_destroy",
    "sufr_hip_last_error", "sufr_hip_set_stream", "sufr_hip_synchronize", "sufr_hip_set_window", "sufr_hip_set_window_retry", "sufr_hip_set_array_budget", "sufr_hip_window_repairs", "sufr_hip_normalize", "sufr_hip_sort_device_u32",
    "sufr_hip_sort_device_u64", "sufr_hip_stitch_device_u32", "sufr_hip_stitch_device_u64", "sufr_hip_build_u32", "sufr_hip_build_u64", "sufr_hip_lcp_pair",
    "sufr_read_sequence_file", "sufr_sequence_data_free", "sufr_write_file", "sufr_hip_create_file", "sufr_hip_create_from_sequence",
    "sufr_hip_shard_build", "sufr_write_frame", "sufr_hip_shard_write", "sufr_hip_create_from_sequence_multi",
    "sufr_hip_create_file_multi",
]
# every symbol include/sufr_query.h declares
QUERY_EXPORTS = [
    "sufr_file_open", "sufr_file_close", "sufr_file_metadata", "sufr_file_text", "sufr_file_seed_mask",
    "sufr_file_suffix_array", "sufr_file_lcp_array", "sufr_file_suffix", "sufr_file_lcp", "sufr_file_sequence_start",
    "sufr_file_sequence_name", "sufr_file_sequence_of", "sufr_file_search", "sufr_file_search_batch",
    "sufr_hip_index_load", "sufr_hip_index_wrap", "sufr_hip_index_free", "sufr_hip_index_width", "sufr_hip_search_batch",
    "sufr_hip_search_batch_device", "sufr_hip_locate_batch_device",
]


class FileMeta(C.Structure):
    """sufr_file_meta"""
    _fields_ = [("version", C.c_uint8), ("is_dna", C.c_uint8), ("allow_ambiguity", C.c_uint8),
                ("ignore_softmask", C.c_uint8), ("index_width", C.c_int),
                ("text_len", C.c_uint64), ("text_pos", C.c_uint64), ("suffix_array_pos", C.c_uint64),
                ("lcp_pos", C.c_uint64), ("len_suffixes", C.c_uint64), ("max_query_len", C.c_uint64),
                ("num_sequences", C.c_uint64), ("seed_mask_len", C.c_uint64), ("file_size", C.c_uint64),
                ("modified", C.c_int64)]

_lib = None


def build_extension(verbose: bool = False) -> Path:
    """Compile the HIP library and the CLI for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", str(CSRC)] + (["asan"] if HOST_ASAN else []), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libsufr_hip.so failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(r.stdout)
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise SufrHipError(-2, f"{LIB_PATH} is missing: build it with sufr_amd.build_extension() "
                               "(there is no CPU fallback)")
    # PyTorch wheels bundle their own HIP runtime; a process that loads this library first (/opt/rocm's runtime) and
    # torch afterwards ends up with two runtimes, and the second one to initialise finds no device.  Whoever uses the
    # Python binding next to torch gets torch's runtime for both: import it first when it is installed.
    import sys
    if "torch" not in sys.modules and not HOST_ASAN:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(str(LIB_PATH))
    vp, u64, u32, cp = C.c_void_p, C.c_uint64, C.c_uint32, C.c_char_p
    L.sufr_hip_abi_version.restype = C.c_int
    L.sufr_hip_device_count.restype = C.c_int
    L.sufr_hip_create.argtypes = [C.c_int]; L.sufr_hip_create.restype = vp
    L.sufr_hip_destroy.argtypes = [vp]; L.sufr_hip_destroy.restype = None
    L.sufr_hip_last_error.argtypes = [vp]; L.sufr_hip_last_error.restype = cp
    L.sufr_hip_set_stream.argtypes = [vp, vp]; L.sufr_hip_set_stream.restype = C.c_int
    L.sufr_hip_synchronize.argtypes = [vp]; L.sufr_hip_synchronize.restype = C.c_int
    L.sufr_hip_set_window.argtypes = [vp, u64, u64]; L.sufr_hip_set_window.restype = C.c_int
    L.sufr_hip_set_window_retry.argtypes = [vp, u64]; L.sufr_hip_set_window_retry.restype = C.c_int
    L.sufr_hip_set_array_budget.argtypes = [vp, u64]; L.sufr_hip_set_array_budget.restype = C.c_int
    L.sufr_hip_window_repairs.argtypes = [vp]; L.sufr_hip_window_repairs.restype = u64
    L.sufr_hip_normalize.argtypes = [vp, vp, u64, C.c_int]; L.sufr_hip_normalize.restype = C.c_int
    dev_sig = [vp, vp, u64, u32, u64, cp, u64, u64, u32, u32, vp, vp, u64, C.POINTER(u64), C.POINTER(Stats)]
    L.sufr_hip_sort_device_u32.argtypes = dev_sig; L.sufr_hip_sort_device_u32.restype = C.c_int
    L.sufr_hip_sort_device_u64.argtypes = dev_sig; L.sufr_hip_sort_device_u64.restype = C.c_int
    host_sig = [vp, vp, u64, u32, u64, cp, u64, u64, vp, vp, vp, u64, C.POINTER(u64), C.POINTER(Stats)]
    L.sufr_hip_build_u32.argtypes = host_sig; L.sufr_hip_build_u32.restype = C.c_int
    L.sufr_hip_build_u64.argtypes = host_sig; L.sufr_hip_build_u64.restype = C.c_int
    L.sufr_hip_lcp_pair.argtypes = [vp, u64, u64, u64]; L.sufr_hip_lcp_pair.restype = u64
    L.sufr_hip_stitch_device_u32.argtypes = [vp, u64, vp, u32, u32, vp]; L.sufr_hip_stitch_device_u32.restype = C.c_int
    L.sufr_hip_stitch_device_u64.argtypes = [vp, u64, vp, u32, u32, vp]; L.sufr_hip_stitch_device_u64.restype = C.c_int
    L.sufr_read_sequence_file.argtypes = [cp, C.c_uint8, C.POINTER(SequenceData), cp, C.c_size_t]
    L.sufr_read_sequence_file.restype = C.c_int
    L.sufr_sequence_data_free.argtypes = [C.POINTER(SequenceData)]; L.sufr_sequence_data_free.restype = None
    L.sufr_write_file.argtypes = [cp, C.c_int, C.c_int, C.c_int, vp, u64, C.c_int, vp, vp, u64, C.c_int, u64, cp,
                                  vp, u64, C.POINTER(cp), cp, C.c_size_t]
    L.sufr_write_file.restype = C.c_int
    L.sufr_hip_create_file.argtypes = [vp, C.POINTER(CreateArgs), cp, C.c_size_t, C.POINTER(Stats)]
    L.sufr_hip_create_file.restype = C.c_int
    L.sufr_hip_create_from_sequence.argtypes = [vp, C.POINTER(SequenceData), C.POINTER(CreateArgs), cp, C.c_size_t,
                                                C.POINTER(Stats)]
    L.sufr_hip_create_from_sequence.restype = C.c_int
    L.sufr_hip_shard_build.argtypes = [vp, C.POINTER(SequenceData), C.POINTER(CreateArgs), u32, u32,
                                       C.POINTER(ShardInfo), C.POINTER(Stats)]
    L.sufr_hip_shard_build.restype = C.c_int
    L.sufr_write_frame.argtypes = [cp, C.POINTER(SequenceData), C.POINTER(CreateArgs), u64, cp, C.c_size_t]
    L.sufr_write_frame.restype = C.c_int
    L.sufr_hip_shard_write.argtypes = [vp, C.POINTER(SequenceData), C.POINTER(CreateArgs), cp, u64, u64, u64, C.c_int,
                                       u64, C.c_int]
    L.sufr_hip_shard_write.restype = C.c_int
    L.sufr_hip_create_from_sequence_multi.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(SequenceData),
                                                      C.POINTER(CreateArgs), cp, C.c_size_t, C.POINTER(Stats)]
    L.sufr_hip_create_from_sequence_multi.restype = C.c_int
    L.sufr_hip_create_file_multi.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(CreateArgs), cp, C.c_size_t,
                                             C.POINTER(Stats)]
    L.sufr_hip_create_file_multi.restype = C.c_int
    # include/sufr_query.h
    L.sufr_file_open.argtypes = [cp, C.POINTER(vp), cp, C.c_size_t]; L.sufr_file_open.restype = C.c_int
    L.sufr_file_close.argtypes = [vp]; L.sufr_file_close.restype = None
    L.sufr_file_metadata.argtypes = [vp, C.POINTER(FileMeta)]; L.sufr_file_metadata.restype = C.c_int
    for name in ("sufr_file_text", "sufr_file_seed_mask", "sufr_file_suffix_array", "sufr_file_lcp_array"):
        getattr(L, name).argtypes = [vp]; getattr(L, name).restype = vp
    for name in ("sufr_file_suffix", "sufr_file_lcp", "sufr_file_sequence_start", "sufr_file_sequence_of"):
        getattr(L, name).argtypes = [vp, u64]; getattr(L, name).restype = u64
    L.sufr_file_sequence_name.argtypes = [vp, u64]; L.sufr_file_sequence_name.restype = cp
    L.sufr_file_search.argtypes = [vp, cp, C.c_size_t, C.c_int, u64, C.POINTER(u64), C.POINTER(u64)]
    L.sufr_file_search.restype = C.c_int
    L.sufr_file_search_batch.argtypes = [vp, vp, vp, u64, C.c_int, u64, vp, vp, C.c_int]; L.sufr_file_search_batch.restype = C.c_int
    L.sufr_hip_index_load.argtypes = [vp, vp, C.POINTER(vp)]; L.sufr_hip_index_load.restype = C.c_int
    L.sufr_hip_index_wrap.argtypes = [vp, vp, u64, vp, u64, u32, u64, cp, C.POINTER(vp)]; L.sufr_hip_index_wrap.restype = C.c_int
    L.sufr_hip_index_free.argtypes = [vp]; L.sufr_hip_index_free.restype = None
    L.sufr_hip_index_width.argtypes = [vp]; L.sufr_hip_index_width.restype = C.c_int
    L.sufr_hip_search_batch.argtypes = [vp, vp, vp, vp, u64, C.c_int, u64, vp, vp]; L.sufr_hip_search_batch.restype = C.c_int
    L.sufr_hip_search_batch_device.argtypes = [vp, vp, vp, vp, u64, C.c_int, u64, vp, vp]
    L.sufr_hip_search_batch_device.restype = C.c_int
    L.sufr_hip_locate_batch_device.argtypes = [vp, vp, vp, vp, u64, u64, vp, vp, u64, C.POINTER(u64)]
    L.sufr_hip_locate_batch_device.restype = C.c_int
    _lib = L
    return L


class Context:
    """sufr_hip_ctx: one per GPU."""

    def __init__(self, device: int = 0):
        L = lib()
        self._h = L.sufr_hip_create(device)
        if not self._h:
            raise SufrHipError(-2, L.sufr_hip_last_error(None).decode())
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            lib().sufr_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int):
        if rc != 0:
            raise SufrHipError(rc, lib().sufr_hip_last_error(self._h).decode())

    def set_window(self, window: int = 0, margin: int = 0):
        """Windowed builds (include/sufr_hip.h, sufr_hip_set_window); 0, 0 restores the defaults."""
        self.check(lib().sufr_hip_set_window(self._h, window, margin))

    def set_window_retry(self, widest_margin: int = 0):
        """Cap of the margin a window is re-built with when a repeat crosses its end (0: what 32 bits allow)."""
        self.check(lib().sufr_hip_set_window_retry(self._h, widest_margin))

    def set_array_budget(self, nbytes: int = 0):
        """Out-of-core windowed create: device bytes the SA + LCP arrays may take at once (sufr_hip_set_array_budget; 0: no limit)."""
        self.check(lib().sufr_hip_set_array_budget(self._h, nbytes))

    @property
    def window_repairs(self) -> int:
        """Suffixes of the last windowed build that were ordered by whole-text comparison."""
        return int(lib().sufr_hip_window_repairs(self._h))

    def synchronize(self):
        self.check(lib().sufr_hip_synchronize(self._h))

    @property
    def handle(self):
        return self._h
