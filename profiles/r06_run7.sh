cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6e
timeout 900 python -m pytest tests/test_gpu_exceptions.py tests/test_gpu_mql_fast.py -q -m gpu -x > gpurun_out/r6e/tests.log 2>&1
tail -3 gpurun_out/r6e/tests.log
python profiles/input_classes.py human human_iupac human_hash > gpurun_out/r6e/classes.txt 2>&1
grep human gpurun_out/r6e/classes.txt
python bench.py > gpurun_out/r6e/bench.json 2> gpurun_out/r6e/bench.err
cat gpurun_out/r6e/bench.json
