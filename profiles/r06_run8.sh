cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6f
for T in 6 8 10 16; do echo "== SUFR_COPY_THREADS=$T"; SUFR_COPY_THREADS=$T python profiles/host_abi_probe.py 3100000000 3 2>&1 | grep -v amdgpu; done | tee gpurun_out/r6f/host_abi2.txt
