cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d
timeout 1500 python -m pytest tests/test_gpu_mql_fast.py -q -m gpu > gpurun_out/r6d/test_mqlfast.log 2>&1
tail -15 gpurun_out/r6d/test_mqlfast.log
python profiles/modes_bench.py "human --dna" > gpurun_out/r6d/modes_human.txt 2>&1
grep -v amdgpu.ids gpurun_out/r6d/modes_human.txt
