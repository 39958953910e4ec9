cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6s
timeout 700 python profiles/soak_round6.py 8 6061 > gpurun_out/r6s/soak_round6.txt 2>&1; tail -3 gpurun_out/r6s/soak_round6.txt
timeout 500 python profiles/soak.py 5 6062 > gpurun_out/r6s/soak.txt 2>&1; tail -2 gpurun_out/r6s/soak.txt
timeout 300 python profiles/soak_families.py 2 6063 > gpurun_out/r6s/soak_families.txt 2>&1; tail -2 gpurun_out/r6s/soak_families.txt
timeout 300 python profiles/soak_runs.py 2 6064 > gpurun_out/r6s/soak_runs.txt 2>&1; tail -2 gpurun_out/r6s/soak_runs.txt
timeout 300 python profiles/soak_wide.py 2 6065 > gpurun_out/r6s/soak_wide.txt 2>&1; tail -2 gpurun_out/r6s/soak_wide.txt
