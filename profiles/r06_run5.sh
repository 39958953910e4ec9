cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d
timeout 1500 python -m pytest tests -q -m gpu -k "max_query_len or mql or capped or options_on or seed_mask_and_max" > gpurun_out/r6d/test_mql.log 2>&1
tail -15 gpurun_out/r6d/test_mql.log
python profiles/modes_bench.py "elegans" > gpurun_out/r6d/modes.txt 2>&1
python profiles/modes_bench.py "ecoli" >> gpurun_out/r6d/modes.txt 2>&1
grep -v amdgpu.ids gpurun_out/r6d/modes.txt
