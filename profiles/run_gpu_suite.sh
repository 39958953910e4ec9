cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6l
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r6l/gputest.log 2>&1
tail -6 gpurun_out/r6l/gputest.log
