#!/bin/bash
# gpurun -- 'bash profiles/search.sh': query tests, then the search rate with a kernel trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/search
timeout 1500 python -m pytest tests/test_gpu_query.py -x -q 2>&1 | tail -15 > gpurun_out/search/tests.txt
timeout 600 python profiles/search_bench.py 1e8 1e7 32 > gpurun_out/search/bench_100m.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/search/prof -o s -- python3 profiles/search_bench.py 3.1e9 1.6e7 32 > gpurun_out/search/bench_3g.txt 2>&1
cat gpurun_out/search/tests.txt gpurun_out/search/bench_100m.txt gpurun_out/search/bench_3g.txt
