#!/bin/bash
# The product's host code under AddressSanitizer + UBSan, the long form (CPU only; nothing here touches a GPU):
#   bash profiles/asan_host.sh [fuzz iterations] > profiles/r04_asan_host.txt
# 1. `make asan` (sufr_io.cpp + sufr_query.cpp + device stubs, g++ -fsanitize=address,undefined);
# 2. tests/test_host_logic.py + tests/test_query.py against that library (libasan preloaded);
# 3. tests/fuzz_host.py: damaged FASTA / FASTQ / gz / bz2 / xz / .sufr inputs, three seeds.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
IT=${1:-4000}
make -C $R/sufr_amd/csrc asan > /dev/null || exit 1
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) SUFR_AMD_HOST_ASAN_LIB=1
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
cd $R
echo "# $(g++ --version | head -1); -fsanitize=address,undefined -fno-sanitize-recover=undefined; $(date -u +%F)"
python3 -m pytest tests/test_host_logic.py tests/test_query.py -q -m "not gpu" -p no:cacheprovider \
    --deselect tests/test_host_logic.py::test_build_fails_loudly_without_gpu \
    --deselect tests/test_host_logic.py::test_no_flat_instructions_in_the_kernels 2>&1 | tail -4
for seed in 7 8 9; do python3 tests/fuzz_host.py $IT $seed 2>&1 | grep -v "SUFR_AMD_HOST_ASAN_LIB is set"; done
