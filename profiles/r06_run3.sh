cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_gpu_exceptions.py -q -m gpu > gpurun_out/r6b/test_exc.log 2>&1
tail -5 gpurun_out/r6b/test_exc.log
bash profiles/kt_class.sh human_iupac > gpurun_out/r6b/kt_iupac.txt 2>&1
grep -i "exc\|human" gpurun_out/r6b/kt_iupac.txt
