cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
timeout 1500 python -m pytest tests/test_gpu_exceptions.py -q -m gpu > gpurun_out/r6b/test_exc.log 2>&1
tail -25 gpurun_out/r6b/test_exc.log
SUFR_HIP_DEBUG=1 python profiles/input_classes.py human human_iupac human_hash > gpurun_out/r6b/iupac.txt 2>&1
grep -v "msd level\|deep level\|  level\|tie level" gpurun_out/r6b/iupac.txt | tail
