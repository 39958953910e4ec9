#!/bin/bash
# the soaks on the final code (one gpurun call):   bash profiles/run_soaks.sh [minutes for soak_round6] [minutes for soak]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
A=${1:-8}; B=${2:-5}
mkdir -p gpurun_out/soaks
timeout $((A * 60 + 200)) python profiles/soak_round6.py $A 7071 > gpurun_out/soaks/soak_round6.txt 2>&1; tail -1 gpurun_out/soaks/soak_round6.txt
timeout $((B * 60 + 200)) python profiles/soak.py $B 7072 > gpurun_out/soaks/soak.txt 2>&1; tail -1 gpurun_out/soaks/soak.txt
timeout 400 python profiles/soak_families.py 3 7073 > gpurun_out/soaks/soak_families.txt 2>&1; tail -1 gpurun_out/soaks/soak_families.txt
timeout 400 python profiles/soak_runs.py 3 7074 > gpurun_out/soaks/soak_runs.txt 2>&1; tail -1 gpurun_out/soaks/soak_runs.txt
timeout 400 python profiles/soak_wide.py 3 7075 > gpurun_out/soaks/soak_wide.txt 2>&1; tail -1 gpurun_out/soaks/soak_wide.txt
timeout 400 python profiles/soak_part.py 2 7076 > gpurun_out/soaks/soak_part.txt 2>&1; tail -1 gpurun_out/soaks/soak_part.txt
