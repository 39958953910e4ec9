#!/bin/bash
# the GPU suite + the driver's smoke on a GPU box:   gpurun -- 'bash profiles/run_gpu_suite.sh'   -> gpurun_out/gputest.log
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -1
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/gputest.log 2>&1
tail -4 gpurun_out/gputest.log
