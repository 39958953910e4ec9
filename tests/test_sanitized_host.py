"""The product's HOST code under AddressSanitizer + UBSan (missing 5 / next 5; SURVEY 5 "Race detection /
sanitizers").  CPU build only: `make asan` in sufr_amd/csrc compiles sufr_io.cpp (parallel FASTA / FASTQ parser; gzip,
bzip2 / xz through hand-declared stream structs; the mmap'd .sufr writer) and sufr_query.cpp (parser of untrusted .sufr
files, host search) with g++ -fsanitize=address,undefined; the device side of the ABI is stubbed (sufr_host_stubs.cpp).
The library is loaded in child processes with libasan preloaded:

* the reader / writer tests of tests/test_host_logic.py and the query tests of tests/test_query.py run against it
  (profiles/asan_host.sh is the long form with more fuzz seeds; its log is profiles/r04_asan_host.txt);
* tests/fuzz_host.py feeds it damaged FASTA / FASTQ / gz / bz2 / xz / .sufr inputs: every one ends in data or in an error
  string, none in a crash or a sanitizer report.
"""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "sufr_amd" / "csrc"


def _asan_env():
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++ for the host-only sanitizer build")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not Path(libasan).exists():
        pytest.skip("libasan.so not installed")
    r = subprocess.run(["make", "-C", str(CSRC), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ)
    env.update(LD_PRELOAD=libasan, SUFR_AMD_HOST_ASAN_LIB="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    return env


def _clean(out: str):
    for bad in ("ERROR: AddressSanitizer", "runtime error:", "SUMMARY: UndefinedBehaviorSanitizer", "Segmentation fault"):
        assert bad not in out, out[-4000:]


def test_sanitizer_build_exports_the_whole_abi():
    """The host-only library answers for every symbol of include/*.h (device entry points: "no device")."""
    env = _asan_env()
    code = ("import sufr_amd; L = sufr_amd.lib(); "
            "[getattr(L, n) for n in sufr_amd.EXPORTS + sufr_amd.QUERY_EXPORTS]; "
            "assert L.sufr_hip_device_count() == 0; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    _clean(r.stdout + r.stderr)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_reader_writer_and_query_tests_pass_under_the_sanitizers():
    env = _asan_env()
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_host_logic.py", "tests/test_query.py", "-q", "-m", "not gpu",
                        "-p", "no:cacheprovider",
                        "--deselect", "tests/test_host_logic.py::test_build_fails_loudly_without_gpu",
                        # (compiles the kernels to ISA with hipcc: nothing of the host code, minutes under a preloaded libasan)
                        "--deselect", "tests/test_host_logic.py::test_no_flat_instructions_in_the_kernels"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    _clean(r.stdout + r.stderr)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]


def test_damaged_inputs_end_in_an_error_string_never_in_a_crash():
    env = _asan_env()
    r = subprocess.run([sys.executable, "tests/fuzz_host.py", "1500", "3"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=1200)
    _clean(r.stdout + r.stderr)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "no crash, no sanitizer report" in r.stdout
