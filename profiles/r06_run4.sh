cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6c
python profiles/input_classes.py human human_iupac human_hash > gpurun_out/r6c/classes.txt 2>&1
cat gpurun_out/r6c/classes.txt | grep human
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r6c/gputest.log 2>&1
tail -8 gpurun_out/r6c/gputest.log
